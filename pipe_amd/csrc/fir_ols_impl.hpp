// Private state of ols::Plan, shared by the two kernel files (fir_ols.hip, fir_ols32.hip).
#pragma once

#include "common.hpp"
#include "fir_ols.hpp"

namespace pipehip {
namespace ols {

struct Plan::Impl {
    DevBuf tw1, tw2;    // fir_ols.hip: W1024^(n1 k2) [16][64], W64^(a d) [4][16]
    DevBuf tw32;        // fir_ols32.hip: W1024^(k n) [32][32]
    DevBuf hperm[2];    // tap spectrum H[0..512] / 1024, double-buffered
    // filters of 513 .. 4096 taps (fir_ols32p.hip): P partitions of Np <= 512 taps, one spectrum each
    int P = 1, Np = 0;
    DevBuf hpart[2];    // [P][513 + 1] spectra, double-buffered like hperm
    DevBuf scratch;     // the partitioned kernel's running sums: [waves][32][64] double2
    AsyncUpload upload; // pinned staging of the spectrum uploads
    int cur = 0;
    int N = 0;
    int cus = 256;
};

// fir_ols32.hip
int init_ols32_tables(Plan::Impl *I);
int run_ols32(const Plan::Impl &I, const void *d_in, int in_dtype, void *d_out, int out_dtype, const double *hist,
              double *hist_new, int64_t frames, int channels, int lines, hipStream_t s, const char **kernel_name,
              KernelTimer *timer);
// fir_ols32p.hip
int run_ols32p(Plan::Impl &I, const void *d_in, int in_dtype, void *d_out, int out_dtype, const double *hist,
               double *hist_new, int64_t frames, int channels, int lines, hipStream_t s, const char **kernel_name,
               KernelTimer *timer);

}  // namespace ols
}  // namespace pipehip
