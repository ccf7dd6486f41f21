// The FIR's history carry, shared by the direct and the overlap-save kernels:
//     new history = last H frames of (old history ++ this call's input)      per Line.
// The history is double-buffered (the kernels read `hist_old`, the next call reads
// `hist_new`), so every main kernel writes the next history itself, spread over the first
// threads of its grid, instead of a follow-up launch.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pipehip {

// THist: float64 everywhere but in the fused chain kernel, which keeps a float32 stream's history
// as float32 (exact, half the bytes: fir.hip converts when a chain changes form).
template <typename TIn, typename THist>
__device__ __forceinline__ void fir_history_carry(const TIn *__restrict__ in, const THist *__restrict__ hist_old,
                                                  THist *__restrict__ hist_new, int64_t frames,
                                                  int64_t line_stride, int H, int C, int lines)
{
    const int64_t per_line = (int64_t)H * C;
    const int64_t total = per_line * lines;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += nthreads) {
        const int64_t line = i / per_line;
        const int r = (int)(i - line * per_line);
        const int j = r / C;
        const int c = r - j * C;
        const int64_t s = frames - H + j;
        const THist v = s >= 0 ? (THist)in[line * line_stride + s * C + c] : hist_old[(line * H + (s + H)) * C + c];
        hist_new[(line * H + j) * C + c] = v;
    }
}

}  // namespace pipehip
