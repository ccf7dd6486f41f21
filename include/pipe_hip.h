/*
 * pipe_hip.h -- C ABI of the MI355X (gfx950) Processor stage bodies for
 * pipelined.dev/pipe.
 *
 * The reference has no FFI: its plugin seam is the Go function-type API
 *     ProcessorAllocatorFunc(mctx, bufferSize, input SignalProperties) (Processor, error)
 *                                                                   line.go:26-30
 *     Processor{mutable.Context, ProcessFunc, StartFunc, FlushFunc, SignalProperties}
 *                                                                   pipe.go:49-60
 * A GPU Processor is an allocator closure whose hooks call the entry points
 * below through cgo (the shim is in INTEGRATION.md).  Each entry point names the
 * reference interface it stands behind.  All functions return a pipe_hip_status
 * (0 = ok); none of them throws, none of them takes or returns a C++/torch type.
 *
 * Threading: goroutines migrate between OS threads and HIP's current device is
 * per thread, so EVERY entry point selects the handle's device itself.  A handle
 * is never entered concurrently by the pipe (run.go:38-52, merger.go:25-30), and
 * different handles share no mutable state.
 *
 * Buffers are interleaved frames x channels (the layout of signal.Floating as
 * the north star defines it); "frames" is what the reference calls Samples /
 * Length (mock.go:43-46,95).
 */
#ifndef PIPE_HIP_H
#define PIPE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIPE_HIP_ABI_VERSION 1

typedef enum pipe_hip_status {
    PIPE_HIP_OK = 0,
    PIPE_HIP_EINVAL = 1,   /* bad argument (allocator error: surfaces from pipe.New, line.go:72-74) */
    PIPE_HIP_ENODEV = 2,   /* no such HIP device / no GPU visible */
    PIPE_HIP_EHIP = 3,     /* a HIP runtime call failed; pipe_hip_last_hip_error() has the code */
    PIPE_HIP_ENOMEM = 4,
    PIPE_HIP_ECAP = 5,     /* output would exceed out_cap_frames (pipe.go:437-443: out is bufferSize) */
    PIPE_HIP_ESTATE = 6,   /* call out of order (e.g. collect without submit; process after a queued launch failed on
                              the device and before the next StartFunc) */
    PIPE_HIP_EBUSY = 7     /* PIPE_HIP_PARAM_RESIDENT: another handle holds this device's doorbell; the handle stays on
                              the plain path (not an error of the stream) */
} pipe_hip_status;

typedef enum pipe_hip_dtype {
    PIPE_HIP_F32 = 0, /* float32 buffers (north-star metric) */
    PIPE_HIP_F64 = 1  /* float64 buffers (what the reference allocates: pipe.go:394,437) */
} pipe_hip_dtype;

/* parameters that a mutation may change between two buffers
 * (mutable.Mutation applied at pipe.go:433, before ProcessFunc) */
typedef enum pipe_hip_param {
    PIPE_HIP_PARAM_GAIN = 0,   /* 1 value                          */
    PIPE_HIP_PARAM_TAPS = 1,   /* ntaps values (count must match)  */
    PIPE_HIP_PARAM_COEFFS = 2, /* nsections*5 values {b0,b1,b2,a1,a2} */
    PIPE_HIP_PARAM_EXACT = 3,  /* 1 value: != 0 pins the stage to its ordered-fma form (bit-exact vs
                                  the oracle) even for float32 results, which otherwise may use the
                                  FIR's overlap-save FFT form / the biquad's time-segmented form /
                                  the fused FIR+biquad+gain chain kernel.  Those "relaxed" forms are
                                  chosen by CALL SIZE (large device-resident batches; for the biquad
                                  also float32 buffers of 1024 frames or more a Line in any call --
                                  thresholds in DESIGN.md), so the same stream can
                                  give different last bits for different call sizes or devices.  Their bound, as tested: the
                                  float64 value differs from the oracle's by O(1e-16) of the filter's
                                  full-scale output, i.e. the float32 result is within one float32
                                  ulp measured at max(|y|, 2^-24 * ||h||_1 * max|x|) -- relative to
                                  the window's full scale, NOT to a quiet sample next to loud ones
                                  (stop-band outputs can be off by many of THEIR ulps).  The
                                  biquad's relaxed forms propagate segment states through powers of
                                  the state-transition matrix, which a resonant section makes
                                  ill-conditioned: their float64 value differs from the oracle's by
                                  the recurrence's own rounding noise -- up to ~200 kappa * 2^-53 of
                                  the Line's full-scale output at single samples, kappa = the
                                  largest entry of any power of the cascade's one-frame transition
                                  matrix (about 1 / sin(w0) for poles at angle w0: 4 for the 1 kHz
                                  Butterworth section, 21 for 300 Hz with Q = 4, at 48 kHz); the
                                  oracle's own distance from the exact result is of that size.
                                  Their float32 result is within one ulp measured at
                                  max(|y|, 2^-19 * kappa * max|y| of the
                                  Line) (tests/test_gpu_biquad_seg.py, scripts/stress_biquad_seg.py).  A cascade with kappa > 1024
                                  or with a section whose poles are not strictly inside the unit
                                  circle always takes the exact form.
                                  float64 buffers always take the exact form.
                                  On a chain it applies to every stage.  The relaxed forms mix
                                  the samples of a 1024-frame window / a segment, so a NaN or Inf
                                  input reaches more outputs than in the ordered form: set this
                                  for streams that may carry non-finite samples. */
    PIPE_HIP_PARAM_RESIDENT = 4, /* 1 value: != 0 keeps the NEXT pipe buffer's work queued on the device ahead of
                                   its call (run.go:198-224's loop, one buffer per pipe_hip_process, moved next
                                   to the data): behind a wait on a doorbell word in pinned host memory sit the
                                   stage's kernels and a store to a completion word.  pipe_hip_process then
                                   copies the buffer into the pinned staging area, rings the doorbell, queues
                                   the work of the call after this one while the device runs, and waits for the
                                   completion word -- no kernel launch and no completion event on the call's
                                   path (per-call figures: DESIGN.md section 5).  Depth per link stays
                                   fitting.go:56-60's: one buffer in the stage at a time, results bit for bit
                                   those of the plain path.
                                   ONE handle per device can hold the doorbell (a queue that waits for a doorbell
                                   costs every other waiting queue of the process tens of microseconds, and a wait
                                   in a shared hardware queue holds up other handles' kernels: measured,
                                   DESIGN.md section 5); the holder's work runs on a stream with a hardware queue of
                                   its own.  A second handle that asks is answered PIPE_HIP_EBUSY and stays on
                                   the plain path; value 0 gives the doorbell back (so does pipe_hip_destroy).
                                   Stages whose state a queued launch cannot be taken back from (the resampler,
                                   biquads of more than two sections) and handles of many Lines answer
                                   PIPE_HIP_EINVAL; a biquad is queued ahead for the calls that take its tile
                                   form (float32 buffers of 1024 frames or more) and runs the plain path for
                                   the others.
                                   Costs.  Idle: the doorbell queue's command processor polls one word; no
                                   compute unit is held.  Busy: the calling thread spins on the completion word
                                   for the length of the stage's kernels (pause, then yield after 50 us, then
                                   50 us sleeps after 2 ms) -- one host core per call in flight, as with the
                                   plain path's completion word.  The queued work assumes the frame count of the
                                   last call: a call that brings another count, a parameter mutation or any
                                   other entry on the handle first runs the queued work on stale input and
                                   drops it (one wasted round trip).  A queue that waits for a doorbell holds up
                                   every device-wide wait of the process (hipDeviceSynchronize, hipFree, work on
                                   the null stream), so queued work is also dropped -- by a watchdog thread of
                                   the library, which only rings and never waits -- when no call has come for
                                   250 ms (a value above 1: that many milliseconds), and by pipe_hip_destroy of
                                   any handle of the device.  Dropped launches are counted: pipe_hip_resident_info.
                                   A queued launch that fails on the device (a look-back wait that gives up, see
                                   PIPE_HIP_PARAM_DEBUG) cannot be run again in another form as on the plain path
                                   -- its successor is already queued on its state: pipe_hip_process answers
                                   PIPE_HIP_EHIP (a ProcessFunc error ends the run, pipe.go:438-440) and
                                   PIPE_HIP_ESTATE until the next pipe_hip_start. */
    PIPE_HIP_PARAM_RELAXED_F64 = 6, /* 1 value: != 0 lets FLOAT64 buffers -- what the Go pipe carries (pipe.go:394,437) -- take
                                  the biquad's time-segmented (tile) form as well, explicit opt-in per handle
                                  (on a chain: for its biquad stages).  Why: the ordered recurrence is one
                                  wave's dependent chain, 22 ns a frame whatever the chip -- 96 us per
                                  4096 x 2 buffer where a host core needs 10; the tile form is one short
                                  launch (per-call figures: DESIGN.md section 5) and can be queued ahead
                                  (PIPE_HIP_PARAM_RESIDENT).  Price: results are no longer bit for bit the
                                  oracle's float64 -- they differ by the recurrence's own rounding noise, at
                                  most ~200 kappa * 2^-53 of the Line's full-scale output (kappa as under
                                  PIPE_HIP_PARAM_EXACT; tested: |y - oracle| <= 256 kappa 2^-53 max|oracle|,
                                  tests/test_gpu_biquad_seg.py) -- nine decimal orders below a float32 ulp.
                                  PIPE_HIP_PARAM_EXACT wins over it.  Calls of fewer than 1024 frames, cascades
                                  of more than two sections per tile pass and unstable sections keep the
                                  ordered form as for float32.
                                  On a FIR (round 6) the same opt-in lets float64 buffers of LARGE calls -- the
                                  sizes at which float32 buffers take it -- use the overlap-save form: the
                                  transform is float64 either way, only the widths of loads and stores differ,
                                  and a float64 batch runs at the rate HBM allows 16 bytes a sample instead of
                                  the ordered sum's 2 N flops a sample.  Price: |y - oracle| <= 64 * 2^-53 *
                                  ||h||_1 * max|x| (tested, tests/test_gpu_fir_ols.py; measured: below 4 * 2^-53
                                  of it) -- the transform's rounding noise, 29 bits below a float32 ulp of the
                                  filter's full-scale output -- instead of bit for bit.  Per-buffer calls keep
                                  the ordered form (they are latency-bound).  A chain hands the parameter to all
                                  of its stages -- and a FIR -> biquad (-> gain) chain whose FIR and biquad both
                                  carry it takes the FUSED kernel on float64 buffers as it does on float32 ones
                                  (one read and one write of the buffers, 16 bytes a sample); bound as tested
                                  (tests/test_gpu_chain_fused.py): the sum of the two stages' bounds above. */
    PIPE_HIP_PARAM_RESIDENT_SHARED = 7, /* 1 value: != 0 as PIPE_HIP_PARAM_RESIDENT, for SEVERAL handles of one device (up to 16):
                                  they share the device's ONE doorbell queue, each queueing its next buffer's work at
                                  the queue's tail while its current buffer runs.  A hardware queue runs in order, so
                                  this pays exactly when the handles are called in the order they were called last
                                  time and never concurrently -- what pipe.Run's synchronous executor does: all Lines
                                  of a context in one goroutine, round-robin, a Line's stages in order (run.go:37-52,
                                  112-132).  Then every stage of every Line has the doorbell's latency (measured: 10.8
                                  us a call for 1 to 16 handles where the plain path takes 11.7; profiles/
                                  r06_shared_doorbell_queue.txt).  A call that finds other handles' work AHEAD of its
                                  own in the queue rings those doorbells first: that work runs on stale input and is
                                  taken back by its owners at their next call (one wasted launch and one plain-path
                                  call each; counted as dropped_by_entry) -- correct in any order, but in the
                                  asynchronous mode (a goroutine per component, merger.go:25-30: arrival order is
                                  anybody's) it costs 27 us a call, so ask for it in the synchronous mode only.
                                  Calls of the sharing handles serialise on one lock (they are not concurrent in that
                                  mode anyway).  Any other entry on a sharing handle (start, flush, a mutation, a batch
                                  call) first takes back EVERYTHING queued on the device.  Exclusive holder and sharers
                                  exclude each other: PIPE_HIP_EBUSY.  Value 0 leaves the queue.  Results: bit for bit
                                  the plain path's, as with PIPE_HIP_PARAM_RESIDENT. */
    PIPE_HIP_PARAM_DEBUG = 5     /* 2 values {tile, limit_us}: the next launch of a look-back form (fused chain, tile
                                  biquad) fails on demand -- its tiles of that index publish nothing and a wait gives
                                  up after limit_us -- the failure a preempted predecessor tile causes.  A
                                  synchronous entry then puts the state back and runs the call again in a form
                                  without look-back (the result is the stream's, the status OK); an asynchronous
                                  pipe_hip_process_batch reports PIPE_HIP_EHIP at the next synchronous entry with
                                  the stream's state as it was BEFORE the failed batch.  For tests. */
} pipe_hip_param;

/* Opaque Processor handle: the state a Go closure would capture. */
typedef struct pipe_hip_processor pipe_hip_processor;

/* What the allocator receives: bufferSize (line.go:27) and the input
 * SignalProperties (line.go:38-41), plus placement. */
typedef struct pipe_hip_config {
    int32_t device;      /* HIP device ordinal */
    int32_t buffer_size; /* frames per pipe buffer (pipe.go:90,107) */
    int32_t channels;    /* input SignalProperties.Channels */
    int32_t dtype;       /* pipe_hip_dtype of in and out buffers */
    int32_t lines;       /* independent Lines driven by this handle (>= 1); state is per Line,
                            parameters are shared -- "N parallel Lines, same chain"            */
    int32_t max_batch;   /* max consecutive buffers per Line in one pipe_hip_process_batch (>= 1) */
} pipe_hip_config;

/* ---- allocators: the body of a ProcessorAllocatorFunc (line.go:26-30) ------ */
/* y = x * gain */
int pipe_hip_gain_create(const pipe_hip_config *cfg, double gain, pipe_hip_processor **out);
/* FIR, same taps for every channel; 1 <= ntaps <= 4096 (ordered direct form -- on the float64
 * matrix pipe for large calls, whose instruction adds its products in exactly that order: the
 * same bits; large float32 batches take the overlap-save form, above 512 taps partitioned:
 * PIPE_HIP_PARAM_EXACT) */
int pipe_hip_fir_create(const pipe_hip_config *cfg, const double *taps, int32_t ntaps,
                        pipe_hip_processor **out);
/* DF2T biquad cascade; coeffs = nsections x {b0,b1,b2,a1,a2}; 1 <= nsections <= 8 */
int pipe_hip_biquad_create(const pipe_hip_config *cfg, const double *coeffs, int32_t nsections,
                           pipe_hip_processor **out);
/* rational polyphase resampler: proto has up*taps_per_phase taps, phase p uses
 * proto[p + j*up].  Output SampleRate = input * up / down.  An up-sampler cannot
 * live behind ProcessFunc with full buffers: out_cap_frames is
 * checked and PIPE_HIP_ECAP returned, nothing consumed. */
int pipe_hip_resampler_create(const pipe_hip_config *cfg, const double *proto,
                              int32_t taps_per_phase, int32_t up, int32_t down,
                              pipe_hip_processor **out);
/* n-input sum ((in0+in1)+in2)...; 2 <= inputs <= 8.  (The reference's merger.go
 * merges error channels, not signals: SURVEY.md F2.) */
int pipe_hip_mix_create(const pipe_hip_config *cfg, int32_t inputs, pipe_hip_processor **out);
/* A Line's Processors slice (line.go:17) run back to back on the device with
 * float64 intermediates that never leave HBM/LDS -- or, for FIR -> biquad (one or two sections)
 * [-> gain] on large float32 batches, as ONE kernel with no intermediates at all (see PIPE_HIP_PARAM_EXACT
 * for its bound).  Takes ownership of the stages (destroying the chain destroys them).  All stages
 * must share cfg. */
int pipe_hip_chain_create(pipe_hip_processor *const *stages, int32_t n_stages,
                          pipe_hip_processor **out);

/* Output SignalProperties the allocator must return (line.go:75, pipe.go:418):
 * channels, and the SampleRate ratio as up/down. */
int pipe_hip_output_properties(const pipe_hip_processor *p, int32_t *channels,
                               int32_t *rate_up, int32_t *rate_down);

/* ---- hooks -------------------------------------------------------------------- */
/* StartFunc (pipe.go:82-83; called at run.go:64-74,177,201): zero all per-Line
 * state.  A pipe may be started again (pipe_test.go:108-131). */
int pipe_hip_start(pipe_hip_processor *p);
/* StartFunc of Lines [first, first + count) of a handle with cfg.lines > 1 only: a Line that
 * joins a RUNNING batch handle (Pipe.AddLine, pipe.go:260-300; multiLineExecutor.addRoute starts
 * the new Line's components between two passes, run.go:134-145) begins from silence while the
 * other Lines keep their state.  Fixed-rate processors; PIPE_HIP_EINVAL otherwise. */
int pipe_hip_start_lines(pipe_hip_processor *p, int32_t first, int32_t count);
/* FlushFunc (pipe.go:84-86; run.go:54-62): drain everything the handle has queued -- on its own
 * stream and on the caller's stream the last device-resident call named (StartFunc drains the same
 * way before it resets the state). */
int pipe_hip_flush(pipe_hip_processor *p);
/* Releases device and pinned memory.  (Go has no destructor hook; the shim ties
 * it to FlushFunc of the last run or a finalizer.) */
int pipe_hip_destroy(pipe_hip_processor *p);

/* ProcessFunc(in, out signal.Floating) (int, error)   pipe.go:62-64, called at :438.
 * in/out are HOST pointers, lines x frames x channels interleaved, element type
 * cfg.dtype.  in_frames <= buffer_size (short last buffer, pipe.go:404-406).  The
 * call is synchronous: on return `out` holds *out_frames frames per Line (the
 * return value the pipe uses to Slice, pipe.go:441-443) and `in` may be freed by
 * the pipe (pipe.go:431).  in and out must not alias. */
int pipe_hip_process(pipe_hip_processor *p, const void *in, int32_t in_frames, void *out,
                     int32_t out_cap_frames, int32_t *out_frames);
/* One step of MANY Lines through one handle (cfg.lines = L): the batched form of the
 * multiLineExecutor pass (run.go:112-132), which calls ProcessFunc once per Line.  ins[l] /
 * outs[l] are the HOST buffers of Line l (its own pool buffers, not contiguous with the
 * others); in_frames[l] <= buffer_size may differ per Line -- a short LAST buffer, or a short
 * read in the middle of a stream (pipe.go:404-406): every Line's state advances by exactly its
 * own in_frames[l] (the pass is cut into runs of consecutive Lines with equal counts, one launch
 * per run; the usual pass, all live Lines equal, is one launch).  A NULL ins[l] is a Line that
 * has ended: its state is undefined until the next pipe_hip_start.  in_frames[l] == 0 with a
 * buffer is an empty read: that Line is not advanced.  out_frames[l] (optional) receives
 * in_frames[l].  Fixed-rate processors only.  Synchronous. */
int pipe_hip_process_lines(pipe_hip_processor *p, const void *const *ins, const int32_t *in_frames,
                           void *const *outs, int32_t *out_frames);
/* The same step when every ins[l] / outs[l] was allocated with pipe_hip_host_alloc (pinned,
 * device-visible -- what a pool built on it hands out, signal.PoolAllocator pipe.go:490-492):
 * the device gathers the L buffers over PCIe itself and scatters the results back, no host
 * copy and no staging DMA.  Passing pageable memory here is undefined behaviour. */
int pipe_hip_process_lines_pinned(pipe_hip_processor *p, const void *const *ins, const int32_t *in_frames,
                                  void *const *outs, int32_t *out_frames);
/* ProcessFunc for the n-input mix: ins[i] are HOST pointers of `frames` frames. */
int pipe_hip_mix_process(pipe_hip_processor *p, const void *const *ins, int32_t n_inputs,
                         int32_t frames, void *out);

/* Asynchronous form of ProcessFunc, mirroring a link of fitting.Async (a channel of capacity 1
 * plus the message in the receiver's hand, internal/fitting/fitting.go:56-60): submit() stages
 * buffer k and returns while the device works; collect() blocks for the OLDEST buffer in flight.
 * Up to TWO buffers may be in flight per handle (two staging slots): submit(k + 1) before
 * collect(k) overlaps the staging and launch of one buffer with the kernels and transfers of the
 * other.  A third submit, a collect with nothing in flight, and pipe_hip_process /
 * pipe_hip_process_lines while something is in flight return PIPE_HIP_ESTATE. */
int pipe_hip_submit(pipe_hip_processor *p, const void *in, int32_t in_frames);
int pipe_hip_collect(pipe_hip_processor *p, void *out, int32_t out_cap_frames, int32_t *out_frames);

/* A mutable.Mutation body (mutable/mutable.go:40-48) for this component: takes
 * effect for the next buffer and never for one already submitted (pipe.go:433).
 * On a chain the parameter goes to the first stage that accepts it (EXACT: to all). */
int pipe_hip_set_param(pipe_hip_processor *p, int32_t param, const double *values, int32_t count);
/* The same for stage `stage` (0-based) of a chain: needed when two stages take the same
 * parameter (two FIRs with equal tap counts). */
int pipe_hip_chain_set_param(pipe_hip_processor *chain, int32_t stage, int32_t param, const double *values,
                             int32_t count);

/* ---- device-resident batch (multiLineExecutor step, run.go:112-132, widened on
 * the time axis): every Line advances by frames_per_line frames in one launch.
 * d_in/d_out are DEVICE pointers on cfg.device, element (line, frame, ch) at
 * ((line*frames_per_line + frame)*channels + ch); frames_per_line <=
 * buffer_size*max_batch.  Asynchronous on `stream` (a hipStream_t, NULL = the
 * handle's own stream); state carries to the next call exactly as if the frames
 * had been processed buffer by buffer.  For the resampler *out_frames (may be
 * NULL otherwise) receives the frames written per Line and d_out must hold
 * out_cap_frames per Line.  These calls return before the device has run them: a
 * device-side failure (the fused chain kernel giving up on a predecessor tile that
 * never shows up, e.g. another process holding half the CUs) is reported as
 * PIPE_HIP_EHIP by the NEXT synchronous entry on the handle -- pipe_hip_process,
 * pipe_hip_collect, pipe_hip_process_lines[_pinned] or pipe_hip_flush. */
int pipe_hip_process_batch(pipe_hip_processor *p, const void *d_in, void *d_out,
                           int64_t frames_per_line, void *stream);
int pipe_hip_resample_batch(pipe_hip_processor *p, const void *d_in, int64_t in_frames_per_line,
                            void *d_out, int64_t out_cap_frames, int64_t *out_frames, void *stream);
int pipe_hip_mix_batch(pipe_hip_processor *p, const void *const *d_ins, int32_t n_inputs,
                       void *d_out, int64_t frames_per_line, void *stream);

/* ---- measurement -------------------------------------------------------------- */
/* When enabled, every batch launch of the handle's dominant kernel is bracketed
 * by hipEvents on the launch stream; kernel_time() synchronises and returns the
 * accumulated milliseconds and launch count since the last reset. */
int pipe_hip_set_profiling(pipe_hip_processor *p, int32_t enabled);
int pipe_hip_kernel_time(pipe_hip_processor *p, double *total_ms, int64_t *launches, int32_t reset);
/* name of the dominant kernel variant the last batch call launched */
const char *pipe_hip_kernel_name(const pipe_hip_processor *p);
/* PIPE_HIP_PARAM_RESIDENT bookkeeping: *holds_doorbell = 1 while the handle holds its device's doorbell;
 * *dropped_by_watchdog / *dropped_by_entry = queued launches that were run on stale input and taken back because
 * no call came within the idle limit / because another entry (a mutation, a short buffer, start, flush, another
 * handle's destroy) came first.  A host that sees dropped_by_watchdog grow is feeding the stage slower than the
 * idle limit: each drop is one wasted round trip on the next call.  Any pointer may be NULL. */
int pipe_hip_resident_info(pipe_hip_processor *p, int32_t *holds_doorbell, int64_t *dropped_by_watchdog,
                           int64_t *dropped_by_entry);

/* ---- utilities ------------------------------------------------------------------ */
int pipe_hip_abi_version(void);
/* bit 0: the library was built with the A/B-only environment switches (`make AB=1`, libpipe_hip_ab.so); the
 * default build reads only the knobs DESIGN.md lists */
int pipe_hip_build_flags(void);
const char *pipe_hip_strerror(int status);
int pipe_hip_last_hip_error(void);
int pipe_hip_device_count(int32_t *count);
/* pinned host memory for the shim's staging buffers (the PoolAllocator analogue
 * of pipe.go:490-492 on the host side of the DMA) */
int pipe_hip_host_alloc(int64_t bytes, void **ptr);
int pipe_hip_host_free(void *ptr);
/* SplitMix64 synthetic stream written on the device:
 * sample i = ((splitmix64(seed, first_index+i) >> 40) * 2^-23) - 1 */
int pipe_hip_synth_fill(int32_t device, void *d_out, int32_t dtype, uint64_t seed,
                        int64_t first_index, int64_t samples, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIPE_HIP_H */
