import numpy as np, torch
from pipe_amd import processors as P, synth
F=4096
def batch(p, lines, frames, C, dtype):
    x = torch.zeros(lines*frames*C, dtype=torch.float32 if dtype==np.float32 else torch.float64, device="cuda")
    y = torch.empty_like(x); p.process_batch(x, y, frames); torch.cuda.synchronize(); return p.kernel_name()
q1 = synth.biquad_rbj_lowpass(); q3 = np.vstack([synth.biquad_rbj_lowpass(fc=f) for f in (500.,1500.,4000.)])
for name, mk in [
 ("bq S1 lines300 C2 f64", lambda: (P.Biquad(q1, 512, 2, dtype=np.float64, lines=300), 300, 512, 2, np.float64)),
 ("bq S1 lines1 C2 f64", lambda: (P.Biquad(q1, F, 2, dtype=np.float64), 1, F, 2, np.float64)),
 ("bq S3 lines1 C2 f64", lambda: (P.Biquad(q3, F, 2, dtype=np.float64), 1, F, 2, np.float64)),
 ("bq S3 lines3000 C2 f64", lambda: (P.Biquad(q3, 256, 2, dtype=np.float64, lines=3000), 3000, 256, 2, np.float64)),
 ("bq S1 lines64 C16 f32 big", lambda: (P.Biquad(q1, F, 16, dtype=np.float32, lines=64, max_batch=8), 64, 8*F, 16, np.float32)),
]:
    p, L, fr, C, dt = mk(); p.start(); print(name, "->", batch(p, L, fr, C, dt)); p.close()
for C, T, up, down in [(2,24,160,147),(8,24,160,147),(64,24,160,147),(2,20,160,147),(3,24,160,147),(2,24,2,1),(2,24,1,2),(64,32,160,147),(64,48,160,147),(16,64,160,147),(32,40,320,147)]:
    try:
        proto = synth.resampler_proto(up, down, T)
        p = P.Resampler(proto, T, up, down, F, C, dtype=np.float32); p.start()
        x = np.zeros((3000 if up > down else F, C), np.float32); p.process(x); print("rs", C, T, up, down, "->", p.kernel_name()); p.close()
    except Exception as e: print("rs", C, T, up, down, "ERR", e)
