// The row form of the polyphase resampler (resampler_rows.hip): interface between the Resampler handle
// (resampler.hip) and the kernel's translation unit.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace pipehip {
namespace rows {

struct Args {
    const void *in;       // [lines][in_frames][C]
    void *out;            // [lines][out_cap][C]
    const double *hist;   // [lines][T-1][C]: the T-1 frames ahead of the call's input
    const double *rtaps;  // [up][T]: row i = the taps of output i of a period, proto[(i down mod up) + j up], j ascending
    int64_t in_frames, out_frames, out_cap;
    int up, down, lines, T;
    int C, rpb;           // channels (even, 2 .. 128) and rows of a workgroup's block: 64 / (C / 2), lane = (row, pair of channels)
    unsigned pair_rcp;    // ceil(65536 / (C / 2)): lane / (C / 2) by one multiplication
    int row_out, row_in;  // a row: B periods of the phase pattern = B up outputs, B down input frames
    int segs;             // waves of a workgroup (<= 16); wave w computes outputs [seg_b[w], seg_b[w + 1]) of every row
    int seg_b[17];
    int blocks_per_line;  // blocks of rows per Line: ceil(rows that hold outputs of this call / (128 / C))
    int fb0;              // the first row's first input frame, relative to the call's input (<= 0 at a stream's start)
    int ob0;              // the first row's first output, relative to the call's first output (<= 0)
    int in_stride;        // LDS: samples between rows of the input (row_in C + pad: the lanes' frame-t reads spread over the banks)
    int out_off;          // LDS: byte offset of the second buffer of input
    int lds_bytes;
    unsigned piece_magic; // floor(2^32 / pieces of a row) + 1: idx / pieces by one multiplication (idx < 65536)
    unsigned long long *prof;  // PH_RR_PROF builds: [wave][8] s_memtime ticks per phase
    int max_groups;       // workgroups resident on the device: a workgroup walks blocks blockIdx, + gridDim, ...
};

// true when a kernel was launched (float32 in and out, T one of 8 / 12 / 16 / 24); events as for hipExtLaunchKernelGGL
bool launch(const Args &a, int in_f64, int out_f64, hipStream_t s, hipEvent_t ev_a, hipEvent_t ev_b);

}  // namespace rows
}  // namespace pipehip
