/*
 * oracle/pipe_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the per-buffer stage execution of
 * pipelined.dev/pipe in its synchronous (`pipe.Run`) mode:
 *   Source/Processor/Sink.execute          pipe.go:379-471
 *   out-pool geometry                       pipe.go:415-421, 490-492
 *   lineExecutor / multiLineExecutor / run  run.go:37-132, 198-224
 *   sync fitting (one slot + closed flag)   internal/fitting/fitting.go:62-79
 *   bind order / props threading            line.go:62-104
 *   mock.Source / mock.Processor / mock.Sink mock/mock.go:86-105,147-154,180-189
 *
 * Pinned against the reference's own known answers (tests/test_oracle_pipe.py):
 * message / frame counts, identity copy, hook ordering, restart doubling
 * (SURVEY.md 8c items 1-8).  `pipelined.dev/signal v0.10.0` (go.mod:3) is not
 * vendored; only the subset observable at the reference's call sites is
 * restated (interleaved frames, length <= capacity, Slice(0,n), pool keyed by
 * (channels,length,capacity)).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 */
#ifndef PIPE_ORACLE_H
#define PIPE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPIPE_MAX_PROCS 8

/* error codes: 0 = nil, OPIPE_EOF = io.EOF, anything else = a component error */
enum { OPIPE_OK = 0, OPIPE_EOF = -1, OPIPE_ERR_CAP = -2 };

/* processor kinds usable inside an oracle Line */
enum {
    OPIPE_PROC_COPY = 0,   /* mock.Processor: signal.FloatingAsFloating  mock.go:147-154 */
    OPIPE_PROC_GAIN = 1,   /* params[0] = gain */
    OPIPE_PROC_FIR = 2,    /* params = taps, n_params = ntaps */
    OPIPE_PROC_BIQUAD = 3, /* params = nsections x {b0,b1,b2,a1,a2} */
};

/* source kinds */
enum {
    OPIPE_SRC_CONST = 0, /* mock.Source: every scalar sample == value  mock.go:100-102 */
    OPIPE_SRC_SYNTH = 1, /* SplitMix64 stream, seed in `seed` */
    OPIPE_SRC_ARRAY = 2, /* user data (interleaved, limit frames) */
};

typedef struct {
    int kind;
    const double *params;
    int n_params;
    int err_on_call;  /* mock.Processor.ErrorOnCall */
    int err_on_start; /* mock.Starter.ErrorOnStart  */
    int err_on_flush; /* mock.Flusher.ErrorOnFlush  */
} opipe_proc_desc;

typedef struct {
    /* mock.Source fields (mock/mock.go:61-72) */
    int src_kind;
    int64_t src_limit; /* frames */
    double src_value;
    int src_channels;
    uint64_t src_seed;
    const double *src_data;
    int src_err_on_call, src_err_on_start, src_err_on_flush;
    int n_procs;
    opipe_proc_desc procs[OPIPE_MAX_PROCS];
    /* mock.Sink fields (mock/mock.go:160-168) */
    int sink_discard;
    int sink_err_on_call, sink_err_on_start, sink_err_on_flush;
} opipe_line_desc;

/* mock.Counter + Starter + Flusher (mock/mock.go:15-58) */
typedef struct {
    int64_t messages;
    int64_t samples; /* frames, as in the reference */
    int started, flushed;
} opipe_counter;

typedef struct {
    opipe_counter source, procs[OPIPE_MAX_PROCS], sink;
    double *sink_values;    /* mock.Sink Counter.Values when !discard; malloc'd */
    int64_t sink_values_len; /* scalar samples */
} opipe_line_result;

typedef struct {
    int err_start; /* "error starting"           run.go:201-203 */
    int err_exec;  /* ErrorRun.ErrExec (0 if EOF) run.go:215-222 */
    int err_flush; /* ErrorRun.ErrFlush           run.go:204-213 */
} opipe_run_error;

typedef struct opipe_pipe opipe_pipe;

/* bind (line.go:62-104): allocate components and connect them with sync
 * fittings and out pools */
opipe_pipe *opipe_bind(int buffer_size, int n_lines, const opipe_line_desc *lines);
/* pipe.Run body (run.go:198-224) on an already bound pipe; may be called again
 * after resetting sources (TestReset, pipe_test.go:108-131) */
void opipe_run(opipe_pipe *p, opipe_run_error *err);
/* mock.Source.Reset mutation (mock.go:111-118) for line i */
void opipe_reset_source(opipe_pipe *p, int line);
/* copies counters out; sink_values pointers stay owned by the pipe */
void opipe_results(opipe_pipe *p, opipe_line_result *results);
void opipe_free(opipe_pipe *p);

/* one-shot helper: bind + run + results; caller frees sink_values with
 * opipe_free_values */
int opipe_run_lines(int buffer_size, int n_lines, const opipe_line_desc *lines,
                    opipe_line_result *results, opipe_run_error *err);
void opipe_free_values(double *values);

#ifdef __cplusplus
}
#endif
#endif
