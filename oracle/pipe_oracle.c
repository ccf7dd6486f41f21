/*
 * oracle/pipe_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Restatement of the reference's synchronous buffer loop; see pipe_oracle.h for
 * the file:line map.  Per-sample access goes through non-inlined accessor
 * functions on purpose: the reference reaches every sample through the
 * signal.Floating interface (mock.go:100-102), which Go neither inlines nor
 * vectorises, and the CPU baseline (oracle/cpu_baseline.c) times this code.
 */
#include "pipe_oracle.h"
#include "dsp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* signal.Floating subset (pipelined.dev/signal v0.10.0, un-vendored; call     */
/* sites: pipe.go:394,401,404-405,431,437,442,447,464; mock.go:95,101,151,185) */
/* ------------------------------------------------------------------------- */
typedef struct obuf {
    double *data; /* interleaved frames */
    int channels;
    int length;   /* frames */
    int capacity; /* frames */
    struct obuf *next_free;
} obuf;

__attribute__((noinline)) static double obuf_sample(const obuf *b, int64_t i)
{
    return b->data[i];
}
__attribute__((noinline)) static void obuf_set_sample(obuf *b, int64_t i, double v)
{
    b->data[i] = v;
}
static int obuf_len(const obuf *b) { return b->length * b->channels; } /* Len()  */

/* signal.GetPoolAllocator(channels, length, capacity)   pipe.go:490-492 */
typedef struct {
    int channels, length, capacity;
    obuf *free_list;
    int64_t allocated; /* buffers ever created: steady state must not grow */
} opool;

static obuf *pool_get(opool *p) /* PoolAllocator.Float64()  pipe.go:394,437 */
{
    obuf *b = p->free_list;
    if (b) {
        p->free_list = b->next_free;
    } else {
        b = calloc(1, sizeof *b);
        b->data = malloc(sizeof(double) * (size_t)p->capacity * (size_t)p->channels + 8);
        b->channels = p->channels;
        b->capacity = p->capacity;
        p->allocated++;
    }
    b->length = p->length; /* recycled contents are NOT cleared */
    b->next_free = NULL;
    return b;
}

static void pool_put(opool *p, obuf *b) /* Floating.Free(pool)  pipe.go:401,431,447,464 */
{
    if (!b || !p)
        return;
    b->next_free = p->free_list;
    p->free_list = b;
}

static void pool_destroy(opool *p)
{
    obuf *b = p->free_list;
    while (b) {
        obuf *n = b->next_free;
        free(b->data);
        free(b);
        b = n;
    }
    p->free_list = NULL;
}

/* ------------------------------------------------------------------------- */
/* syncFitting                                   internal/fitting/fitting.go:62-79 */
/* ------------------------------------------------------------------------- */
typedef struct {
    int closed;
    obuf *message; /* Message.Signal; Mutations are nil in pipe.Run (pipe.go:100) */
} ofitting;

static int fitting_send(ofitting *f, obuf *m) /* :62-68 */
{
    if (f->closed)
        return 0;
    f->message = m;
    return 1;
}
static obuf *fitting_receive(ofitting *f, int *ok) /* :70-75 -- stale msg + false when closed */
{
    *ok = !f->closed;
    return f->message;
}
static void fitting_close(ofitting *f) { f->closed = 1; } /* :77-79 */

/* ------------------------------------------------------------------------- */
/* components                                                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    ofitting fitting; /* out.sender    line.go:51-54 */
    opool pool;       /* out.allocator */
} olink;

typedef struct ocomp ocomp;
typedef int (*exec_fn)(ocomp *);
struct ocomp {
    exec_fn execute;
    int role; /* 0 source, 1 processor, 2 sink */
    int channels_out; /* SignalProperties.Channels of this stage's output */
    olink out;        /* unused for sinks */
    olink *in;        /* upstream out (in.insert, line.go:155-158) */
    opipe_counter counter;
    int err_on_call, err_on_start, err_on_flush;
    /* source */
    int src_kind;
    int64_t limit;
    double value;
    uint64_t seed;
    const double *src_data;
    /* processor */
    int proc_kind;
    double gain;
    odsp_fir *fir;
    odsp_biquad *biquad;
    double *scratch_in, *scratch_out; /* contiguous staging for the DSP bodies */
    /* sink */
    int discard;
    double *values;
    int64_t values_len, values_cap;
};

/* mock.Source SourceFunc                                   mock/mock.go:86-105 */
static int source_func(ocomp *s, obuf *out, int *read_out)
{
    if (s->err_on_call)
        return s->err_on_call;
    if (s->counter.samples == s->limit)
        return OPIPE_EOF;
    int64_t read = out->length;
    int64_t left = s->limit - s->counter.samples;
    if (left < read)
        read = left;
    const int C = s->channels_out;
    const int64_t base = s->counter.samples * C;
    for (int64_t i = 0; i < read * C; i++) {
        double v;
        if (s->src_kind == OPIPE_SRC_CONST) {
            v = s->value; /* flat index: every channel gets Value  mock.go:100-102 */
        } else if (s->src_kind == OPIPE_SRC_SYNTH) {
            odsp_synth_fill(s->seed, base + i, &v, 1);
        } else {
            v = s->src_data[base + i];
        }
        obuf_set_sample(out, i, v);
    }
    s->counter.messages++; /* Counter.advance(read)  mock.go:43-46,103 */
    s->counter.samples += read;
    *read_out = (int)read;
    return OPIPE_OK;
}

/* Source.execute                                               pipe.go:379-413 */
static int source_execute(ocomp *s)
{
    /* mutation poll / ctx.Done: dest is nil under pipe.Run (pipe.go:100,383-392) */
    obuf *output = pool_get(&s->out.pool); /* :394 */
    int read = 0;
    int err = source_func(s, output, &read);
    if (err) { /* :399-403 */
        fitting_close(&s->out.fitting);
        pool_put(&s->out.pool, output);
        return err;
    }
    if (read != output->length) /* :404-406 Slice(0, read) */
        output->length = read;
    if (!fitting_send(&s->out.fitting, output)) { /* :408-411 */
        fitting_close(&s->out.fitting);
        return OPIPE_EOF;
    }
    return OPIPE_OK;
}

/* the ProcessFunc bodies.  COPY restates signal.FloatingAsFloating as used by
 * mock.Processor (mock.go:147-154): copy min(len) samples, return frames. */
static int process_func(ocomp *p, const obuf *in, obuf *out, int *processed)
{
    if (p->err_on_call)
        return p->err_on_call;
    int64_t n = obuf_len(in) < obuf_len(out) ? obuf_len(in) : obuf_len(out);
    const int C = in->channels;
    const int frames = (int)(n / C);
    switch (p->proc_kind) {
    case OPIPE_PROC_COPY:
        for (int64_t i = 0; i < n; i++)
            obuf_set_sample(out, i, obuf_sample(in, i));
        break;
    case OPIPE_PROC_GAIN:
        for (int64_t i = 0; i < n; i++)
            obuf_set_sample(out, i, obuf_sample(in, i) * p->gain);
        break;
    case OPIPE_PROC_FIR:
    case OPIPE_PROC_BIQUAD:
        /* in must not be retained (pipe.go:431): the DSP state copies what it
         * needs.  Samples travel through the accessors like any Go Processor. */
        for (int64_t i = 0; i < n; i++)
            p->scratch_in[i] = obuf_sample(in, i);
        if (p->proc_kind == OPIPE_PROC_FIR)
            odsp_fir_process(p->fir, p->scratch_in, p->scratch_out, frames);
        else
            odsp_biquad_process(p->biquad, p->scratch_in, p->scratch_out, frames);
        for (int64_t i = 0; i < n; i++)
            obuf_set_sample(out, i, p->scratch_out[i]);
        break;
    default:
        return 1000;
    }
    p->counter.messages++; /* mock.go:152 */
    p->counter.samples += frames;
    *processed = frames;
    return OPIPE_OK;
}

/* Processor.execute                                            pipe.go:423-451 */
static int processor_execute(ocomp *p)
{
    int ok;
    obuf *m = fitting_receive(&p->in->fitting, &ok); /* :426 */
    if (!ok) {
        fitting_close(&p->out.fitting); /* :428 */
        return OPIPE_EOF;
    }
    /* Mutations.ApplyTo: nil map fast path (:433) */
    obuf *output = pool_get(&p->out.pool); /* :437 */
    int processed = 0;
    int err = process_func(p, m, output, &processed);
    int ret = OPIPE_OK;
    if (err) { /* :438-440 -- output is not freed on this path */
        fitting_close(&p->out.fitting);
        ret = err;
    } else {
        if (processed != p->out.pool.length) /* :441-443 */
            output->length = processed;
        if (!fitting_send(&p->out.fitting, output)) { /* :445-449 */
            fitting_close(&p->out.fitting);
            pool_put(&p->out.pool, output);
            ret = OPIPE_EOF;
        }
    }
    pool_put(&p->in->pool, m); /* deferred Free to the UPSTREAM pool (:431) */
    return ret;
}

/* mock.Sink SinkFunc                                       mock/mock.go:180-189 */
static int sink_func(ocomp *s, const obuf *in)
{
    if (s->err_on_call)
        return s->err_on_call;
    if (!s->discard) { /* Values.Append(in) */
        int64_t n = obuf_len(in);
        if (s->values_len + n > s->values_cap) {
            int64_t cap = s->values_cap ? s->values_cap * 2 : 1024;
            while (cap < s->values_len + n)
                cap *= 2;
            s->values = realloc(s->values, sizeof(double) * (size_t)cap);
            s->values_cap = cap;
        }
        for (int64_t i = 0; i < n; i++)
            s->values[s->values_len + i] = obuf_sample(in, i);
        s->values_len += n;
    }
    s->counter.messages++;
    s->counter.samples += in->length; /* advance(in.Length())  mock.go:187 */
    return OPIPE_OK;
}

/* Sink.execute                                                 pipe.go:457-471 */
static int sink_execute(ocomp *s)
{
    int ok;
    obuf *m = fitting_receive(&s->in->fitting, &ok);
    if (!ok)
        return OPIPE_EOF;
    int err = sink_func(s, m);
    pool_put(&s->in->pool, m); /* deferred Free (:464) */
    return err;
}

/* hooks: mock.Starter.Start / mock.Flusher.Flush           mock/mock.go:48-58 */
static int comp_start(ocomp *c)
{
    c->counter.started = 1;
    if (c->role == 1) { /* a re-started pipe must see fresh DSP state */
        if (c->fir)
            odsp_fir_reset(c->fir);
        if (c->biquad)
            odsp_biquad_reset(c->biquad);
    }
    return c->err_on_start;
}
static int comp_flush(ocomp *c)
{
    c->counter.flushed = 1;
    return c->err_on_flush;
}

/* ------------------------------------------------------------------------- */
/* executors                                                   run.go:20-132  */
/* ------------------------------------------------------------------------- */
typedef struct {
    int route;
    int started;
    int n;
    ocomp **executors;
} oline_exec;

/* lineExecutor.execute                                          run.go:37-52 */
static int line_execute(oline_exec *le)
{
    int err = OPIPE_OK;
    for (int i = 0; i < le->started; i++) {
        err = le->executors[i]->execute(le->executors[i]);
        if (err == OPIPE_OK)
            continue;
        if (err == OPIPE_EOF)
            continue; /* keep going so that EOF propagates downstream */
        return err;
    }
    return err;
}

/* lineExecutor.flushHook                                        run.go:54-62 */
static int line_flush(oline_exec *le)
{
    int first = 0;
    for (int i = 0; i < le->started; i++) {
        int e = comp_flush(le->executors[i]);
        if (e && !first)
            first = e;
    }
    return first;
}

/* lineExecutor.startHook                                        run.go:64-74 */
static int line_start(oline_exec *le)
{
    for (int i = 0; i < le->n; i++) {
        int e = comp_start(le->executors[i]);
        if (e)
            return e;
        le->started++;
    }
    return 0;
}

struct opipe_pipe {
    int buffer_size;
    int n_lines;
    opipe_line_desc *descs;
    ocomp **comps;   /* per line: 2 + n_procs components */
    int *n_comps;
    /* multiLineExecutor.executors (run.go:30-34): live lines */
    oline_exec *lines;  /* one per route, fixed order */
    oline_exec **live;  /* the executor's current slice */
    int n_live;
};

/* multiLineExecutor.flushHook                                 run.go:101-110 */
static int multi_flush(opipe_pipe *p)
{
    int first = 0;
    for (int i = 0; i < p->n_live; i++) {
        int e = line_flush(p->live[i]);
        if (e && !first)
            first = e;
    }
    return first;
}

/* multiLineExecutor.startHook                                   run.go:76-99 */
static int multi_start(opipe_pipe *p, int *flush_err)
{
    int start_err = 0;
    for (int i = 0; i < p->n_live; i++) {
        int e = line_start(p->live[i]);
        if (e) {
            start_err = e;
            break;
        }
    }
    if (!start_err)
        return 0;
    *flush_err = multi_flush(p); /* flush what did start (:94) */
    return start_err;
}

/* multiLineExecutor.execute                                   run.go:112-132 */
static int multi_execute(opipe_pipe *p)
{
    int err = OPIPE_OK;
    for (int i = 0; i < p->n_live;) {
        err = line_execute(p->live[i]);
        if (err == OPIPE_OK) {
            i++;
            continue;
        }
        if (err == OPIPE_EOF) {
            int fe = line_flush(p->live[i]); /* :121 */
            if (fe)
                return fe;
            memmove(&p->live[i], &p->live[i + 1],
                    sizeof(oline_exec *) * (size_t)(p->n_live - i - 1)); /* :124 */
            p->n_live--;
            if (p->n_live > 0)
                continue;
        }
        return err;
    }
    return OPIPE_OK;
}

/* ------------------------------------------------------------------------- */
/* bind                                                      line.go:62-118    */
/* ------------------------------------------------------------------------- */
static void link_connect(olink *l, int channels, int buffer_size)
{
    pool_destroy(&l->pool);
    memset(l, 0, sizeof *l);
    l->pool.channels = channels;
    l->pool.length = buffer_size;   /* GetPoolAllocator(ch, bufferSize, bufferSize) */
    l->pool.capacity = buffer_size; /* pipe.go:490-492 */
}

opipe_pipe *opipe_bind(int buffer_size, int n_lines, const opipe_line_desc *lines)
{
    opipe_pipe *p = calloc(1, sizeof *p);
    p->buffer_size = buffer_size;
    p->n_lines = n_lines;
    p->descs = malloc(sizeof(opipe_line_desc) * (size_t)n_lines);
    memcpy(p->descs, lines, sizeof(opipe_line_desc) * (size_t)n_lines);
    p->comps = calloc((size_t)n_lines, sizeof(ocomp *));
    p->n_comps = calloc((size_t)n_lines, sizeof(int));
    p->lines = calloc((size_t)n_lines, sizeof(oline_exec));
    p->live = calloc((size_t)n_lines + 1, sizeof(oline_exec *));
    for (int i = 0; i < n_lines; i++) {
        const opipe_line_desc *d = &lines[i];
        const int n = 2 + d->n_procs;
        ocomp *c = calloc((size_t)n, sizeof(ocomp));
        p->comps[i] = c;
        p->n_comps[i] = n;
        /* source allocator (line.go:63-67, mock.go:79-85) */
        c[0].execute = source_execute;
        c[0].role = 0;
        c[0].channels_out = d->src_channels;
        c[0].src_kind = d->src_kind;
        c[0].limit = d->src_limit;
        c[0].value = d->src_value;
        c[0].seed = d->src_seed;
        c[0].src_data = d->src_data;
        c[0].err_on_call = d->src_err_on_call;
        c[0].err_on_start = d->src_err_on_start;
        c[0].err_on_flush = d->src_err_on_flush;
        int prev_channels = d->src_channels; /* prevProps threading (line.go:67,75) */
        for (int k = 0; k < d->n_procs; k++) {
            ocomp *q = &c[1 + k];
            const opipe_proc_desc *pd = &d->procs[k];
            q->execute = processor_execute;
            q->role = 1;
            q->channels_out = prev_channels; /* mock.Processor echoes props (mock.go:144) */
            q->proc_kind = pd->kind;
            q->err_on_call = pd->err_on_call;
            q->err_on_start = pd->err_on_start;
            q->err_on_flush = pd->err_on_flush;
            if (pd->kind == OPIPE_PROC_GAIN)
                q->gain = pd->params[0];
            if (pd->kind == OPIPE_PROC_FIR)
                q->fir = odsp_fir_new(pd->params, pd->n_params, prev_channels);
            if (pd->kind == OPIPE_PROC_BIQUAD)
                q->biquad = odsp_biquad_new(pd->params, pd->n_params / 5, prev_channels);
            if (pd->kind == OPIPE_PROC_FIR || pd->kind == OPIPE_PROC_BIQUAD) {
                size_t n_s = (size_t)buffer_size * (size_t)prev_channels + 1;
                q->scratch_in = malloc(sizeof(double) * n_s);
                q->scratch_out = malloc(sizeof(double) * n_s);
            }
            prev_channels = q->channels_out;
        }
        ocomp *s = &c[n - 1];
        s->execute = sink_execute;
        s->role = 2;
        s->discard = d->sink_discard;
        s->err_on_call = d->sink_err_on_call;
        s->err_on_start = d->sink_err_on_start;
        s->err_on_flush = d->sink_err_on_flush;
    }
    return p;
}

/* route.connect (line.go:92-104) + route.executor (line.go:106-118) */
static void pipe_connect(opipe_pipe *p)
{
    for (int i = 0; i < p->n_lines; i++) {
        ocomp *c = p->comps[i];
        const int n = p->n_comps[i];
        for (int k = 0; k < n - 1; k++) {
            link_connect(&c[k].out, c[k].channels_out, p->buffer_size);
            c[k + 1].in = &c[k].out;
        }
        oline_exec *le = &p->lines[i];
        free(le->executors);
        le->route = i;
        le->started = 0;
        le->n = n;
        le->executors = malloc(sizeof(ocomp *) * (size_t)n);
        for (int k = 0; k < n; k++)
            le->executors[k] = &c[k];
        p->live[i] = le;
    }
    p->n_live = p->n_lines;
}

/* run()                                                       run.go:198-224 */
void opipe_run(opipe_pipe *p, opipe_run_error *err)
{
    memset(err, 0, sizeof *err);
    pipe_connect(p);
    int flush_err = 0;
    int start_err = multi_start(p, &flush_err);
    if (start_err) {
        err->err_start = start_err;
        err->err_flush = flush_err;
        return;
    }
    int e = OPIPE_OK;
    while (e == OPIPE_OK)
        e = multi_execute(p);
    err->err_flush = multi_flush(p); /* deferred flushHook (:204-213) */
    err->err_exec = (e == OPIPE_EOF) ? 0 : e;
}

void opipe_reset_source(opipe_pipe *p, int line)
{
    opipe_counter *c = &p->comps[line][0].counter;
    c->messages = 0; /* m.Counter = Counter{}  mock.go:114 */
    c->samples = 0;
}

void opipe_results(opipe_pipe *p, opipe_line_result *results)
{
    for (int i = 0; i < p->n_lines; i++) {
        ocomp *c = p->comps[i];
        const int n = p->n_comps[i];
        memset(&results[i], 0, sizeof results[i]);
        results[i].source = c[0].counter;
        for (int k = 0; k < n - 2; k++)
            results[i].procs[k] = c[1 + k].counter;
        results[i].sink = c[n - 1].counter;
        results[i].sink_values = c[n - 1].values;
        results[i].sink_values_len = c[n - 1].values_len;
    }
}

void opipe_free(opipe_pipe *p)
{
    if (!p)
        return;
    for (int i = 0; i < p->n_lines; i++) {
        ocomp *c = p->comps[i];
        for (int k = 0; k < p->n_comps[i]; k++) {
            pool_destroy(&c[k].out.pool);
            odsp_fir_free(c[k].fir);
            odsp_biquad_free(c[k].biquad);
            free(c[k].scratch_in);
            free(c[k].scratch_out);
            free(c[k].values);
        }
        free(c);
        free(p->lines[i].executors);
    }
    free(p->comps);
    free(p->n_comps);
    free(p->lines);
    free(p->live);
    free(p->descs);
    free(p);
}

int opipe_run_lines(int buffer_size, int n_lines, const opipe_line_desc *lines,
                    opipe_line_result *results, opipe_run_error *err)
{
    opipe_pipe *p = opipe_bind(buffer_size, n_lines, lines);
    if (!p)
        return 1;
    opipe_run(p, err);
    opipe_results(p, results);
    for (int i = 0; i < n_lines; i++) { /* hand the sink captures to the caller */
        ocomp *s = &p->comps[i][p->n_comps[i] - 1];
        s->values = NULL;
    }
    opipe_free(p);
    return 0;
}

void opipe_free_values(double *values) { free(values); }
