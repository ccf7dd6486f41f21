/*
 * pipe_host.h -- C entry points of the host-side mirror of pipe.Run /
 * pipe.New+Start+Wait (pipe_amd/csrc/host), so that the compiled C++ host layer
 * can be driven from the pytest harness exactly like pipe_test.go drives the
 * reference: build Lines out of mock.Source / Processors / mock.Sink
 * descriptions, run them, read the mock counters back.
 *
 * This is a TEST/BENCH harness ABI.  The drop-in boundary for third-party host
 * code is include/pipe_hip.h.
 */
#ifndef PIPE_HOST_H
#define PIPE_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIPE_HOST_MAX_PROCS 8

enum { /* processor kinds */
    PIPE_HOST_PROC_MOCK = 0,       /* mock.Processor, host pass-through (mock.go:139-157) */
    PIPE_HOST_PROC_HIP_COPY = 1,   /* HIP gain(1.0)                                       */
    PIPE_HOST_PROC_HIP_GAIN = 2,   /* params[0] = gain                                    */
    PIPE_HOST_PROC_HIP_FIR = 3,    /* params = taps                                       */
    PIPE_HOST_PROC_HIP_BIQUAD = 4, /* params = nsections x {b0,b1,b2,a1,a2}               */
    PIPE_HOST_PROC_HIP_CHAIN = 5   /* params = {ntaps, taps..., nsections, coeffs..., gain} fused on device */
};
enum { PIPE_HOST_SRC_CONST = 0, PIPE_HOST_SRC_SYNTH = 1, PIPE_HOST_SRC_ARRAY = 2 };
enum {
    PIPE_HOST_MODE_RUN = 0,        /* pipe.Run */
    PIPE_HOST_MODE_ASYNC = 1,      /* pipe.New + Start + Wait */
    PIPE_HOST_MODE_RUN_BATCHED = 2 /* stage-major Run: a HIP_CHAIN with identical parameters at the same
                                      position of every Line advances with ONE launch per pass */
};

typedef struct pipe_host_proc_desc {
    int32_t kind;
    const double *params;
    int32_t n_params;
    int32_t err_on_call, err_on_start, err_on_flush, err_on_make;
    int32_t mutate_gain;     /* HIP_GAIN only: push SetGain(mutated_gain) as a Start initializer */
    double mutated_gain;
    int32_t insert_before_pass; /* RUN_BATCHED only: > 0 = the Line is bound WITHOUT this Processor and it is
                                   inserted while the pipe runs, before that pass (Pipe.InsertProcessor,
                                   pipe.go:302-365) */
} pipe_host_proc_desc;

typedef struct pipe_host_line_desc {
    int32_t src_kind;
    int64_t src_limit; /* frames */
    double src_value;
    int32_t src_channels;
    uint64_t src_seed;
    const double *src_data;
    int32_t src_err_on_call, src_err_on_start, src_err_on_flush, src_err_on_make;
    int32_t n_procs;
    pipe_host_proc_desc procs[PIPE_HOST_MAX_PROCS];
    int32_t sink_discard;
    int32_t sink_err_on_call, sink_err_on_start, sink_err_on_flush, sink_err_on_make;
    int32_t join_before_pass; /* RUN_BATCHED only: > 0 = the Line is not bound at the start but added to the
                                 running pipe before that pass (Pipe.AddLine, pipe.go:260-300) */
} pipe_host_line_desc;

typedef struct pipe_host_counter {
    int64_t messages, samples;
    int32_t started, flushed;
} pipe_host_counter;

typedef struct pipe_host_line_result {
    pipe_host_counter source, procs[PIPE_HOST_MAX_PROCS], sink;
    double *sink_values; /* free with pipe_host_free_values */
    int64_t sink_values_len;
} pipe_host_line_result;

typedef struct pipe_host_error {
    int32_t failed;        /* 0 == nil error */
    int32_t is_mock_error; /* errors.Is(err, mockError) */
    int32_t is_bind_error; /* error came from New/Run binding (line.go:62-90) */
    char message[512];
} pipe_host_error;

/* runs = 1 + number of re-Starts with source.Reset() initializers (async mode;
 * TestReset, pipe_test.go:108-131) */
int pipe_host_run(int32_t mode, int32_t buffer_size, int32_t n_lines, const pipe_host_line_desc *lines,
                  pipe_host_line_result *results, pipe_host_error *err, int32_t runs, int32_t device);
void pipe_host_free_values(double *values);
/* signal buffers created by every PoolAllocator of this process so far (PoolAllocator, pipe.go:490-492):
 * a running pipe recycles its buffers, so the count does not depend on how long the Lines are */
int64_t pipe_host_pool_buffers_created(void);

/* What a binding spends per ProcessFunc call on moving one frames x channels float64 buffer into its staging slice
 * and the result back (pipe_amd/csrc/host/binding_cost.cpp): median microseconds over `reps` buffers,
 *   out_us[0] one interface call per sample (what integration/go/hip/hip.go did until round 4), float64 staging
 *   out_us[1] bulk copies (signal.ReadFloat64 / signal.WriteFloat64, mock/mock_test.go:120,128), float64 staging
 *   out_us[2] / out_us[3] the same two with float32 staging (Options.Float32).
 * staging_in / staging_out: frames x channels doubles each (pinned memory of the caller), or NULL. */
int pipe_host_binding_cost(int32_t frames, int32_t channels, int32_t reps, double *staging_in, double *staging_out,
                           double *out_us);

#ifdef __cplusplus
}
#endif
#endif
