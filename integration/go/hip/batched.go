package hip

// Many Lines, one launch per pass (SURVEY.md 8 f4).
//
// pipe.Run's multiLineExecutor is Line-major (run.go:112-132): Line 0's Source, Processors and
// Sink, then Line 1's ... -- L Lines cost L ProcessFunc calls = L launches per pass.  A Batch is
// ONE device handle with cfg.lines = L behind the Processors of L Lines; with the stage-major
// executor of ../pipe_patch/run_batched.go every pass advances all of them with one
// pipe_hip_process_lines call.  SOURCE ONLY, never compiled (see hip.go).

/*
#include "pipe_hip.h"
*/
import "C"

import (
	"context"
	"errors"
	"runtime"
	"unsafe"

	"pipelined.dev/pipe"
	"pipelined.dev/pipe/mutable"
	"pipelined.dev/signal"
)

// Batch is the shared handle of `lines` Lines running the same fixed-rate chain.
type Batch struct {
	stages []*Stage
	lines  int
	opts   Options

	p                 *C.pipe_hip_processor
	bufferSize, chans int
	mctx              mutable.Context
	// per slot: pinned staging (the pool buffers of signal v0.10.0 are not pinned; a pool built
	// on pipe_hip_host_alloc would let pipe_hip_process_lines_pinned read them in place)
	in, out   [][]float64
	inP, outP []unsafe.Pointer
	slabs     []unsafe.Pointer // the two pinned blocks the rows are carved from
	frames    []C.int32_t
	written   []C.int32_t
	live      []bool
	// StartFunc bookkeeping (the C++ mirror's HipBatch::start): slots started and not flushed, and
	// whether a pass has run since the whole handle was last started
	started int
	dirty   bool
}

// BatchedChain: stages as in Chain; Allocator(slot) is the allocator of Line `slot`.
func BatchedChain(o Options, lines int, stages ...*Stage) *Batch {
	return &Batch{stages: stages, lines: lines, opts: o}
}

// Slots implements pipe.BatchGroup (../pipe_patch/run_batched.go).
func (b *Batch) Slots() int { return b.lines }

// Close releases the shared handle and the slots' pinned buffers.
func (b *Batch) Close() {
	// (a second bind() of the same object sets the finalizer again: Go panics on "finalizer already set"
	// unless the first one has been cleared)
	runtime.SetFinalizer(b, nil)
	if b.p != nil {
		C.pipe_hip_destroy(b.p)
		b.p = nil
	}
	for i := range b.slabs {
		C.pipe_hip_host_free(b.slabs[i])
	}
	b.slabs, b.inP, b.outP, b.in, b.out = nil, nil, nil, nil, nil
}

func (b *Batch) bind(mctx mutable.Context, bufferSize int, in pipe.SignalProperties) error {
	if b.p != nil {
		if bufferSize != b.bufferSize || in.Channels != b.chans {
			return errors.New("hip.Batch: every Line must have the same buffer size and channels")
		}
		return nil
	}
	o := b.opts
	o.Float32 = false // (the batch's rows are the pipe's float64 samples, read and written in place)
	cfg := o.config(bufferSize, in.Channels, b.lines)
	chain, err := Chain(b.opts, b.stages...).create(&cfg)
	if err != nil {
		return err
	}
	b.p, b.bufferSize, b.chans, b.mctx = chain, bufferSize, in.Channels, mctx
	runtime.SetFinalizer(b, (*Batch).Close)
	// ONE pinned block per direction, carved into the slots' rows: rows that lie back to back are moved by the
	// DMA engines, a chunk of Lines per copy, both directions at once (pipe_hip_process_lines_pinned: 39-41 GB/s
	// each way at 512 x 4096 x 8; one allocation per row keeps the row kernels, 24-29 GB/s)
	n := bufferSize * in.Channels
	slabIn, pIn, err := pinned(n * b.lines)
	if err != nil {
		b.Close()
		return err
	}
	b.slabs = append(b.slabs, pIn)
	slabOut, pOut, err := pinned(n * b.lines)
	if err != nil {
		b.Close()
		return err
	}
	b.slabs = append(b.slabs, pOut)
	for i := 0; i < b.lines; i++ {
		b.in, b.inP = append(b.in, slabIn[i*n:(i+1)*n]), append(b.inP, unsafe.Pointer(&slabIn[i*n]))
		b.out, b.outP = append(b.out, slabOut[i*n:(i+1)*n]), append(b.outP, unsafe.Pointer(&slabOut[i*n]))
	}
	b.frames = make([]C.int32_t, b.lines)
	b.written = make([]C.int32_t, b.lines)
	b.live = make([]bool, b.lines)
	return nil
}

// Allocator of Line `slot`.  The returned Processor's ProcessFunc refuses to run: these
// Processors only run under pipe.RunBatched, which calls ProcessLines once per pass.
func (b *Batch) Allocator(slot int) pipe.ProcessorAllocatorFunc {
	return func(mctx mutable.Context, bufferSize int, in pipe.SignalProperties) (pipe.Processor, error) {
		if err := b.bind(mctx, bufferSize, in); err != nil {
			return pipe.Processor{}, err
		}
		return pipe.Processor{
			SignalProperties: in, // a fixed-rate chain keeps rate and channels
			// The Lines of a group start together before the first pass (run.go:76-85): the first of
			// them starts the WHOLE handle once, the others find it started.  A Line added to a
			// running pipe (pipe.go:260-300) must not reset the Lines already streaming through the
			// same handle: its slot alone starts from silence.
			StartFunc: func(context.Context) error {
				var st C.int
				switch {
				case b.started == 0:
					st = C.pipe_hip_start(b.p)
					b.dirty = false
				case b.dirty:
					st = C.pipe_hip_start_lines(b.p, C.int32_t(slot), 1)
				}
				if err := status(st, "start"); err != nil {
					return err // (a failed StartFunc gets no FlushFunc: run.go:54-62)
				}
				b.started++
				return nil
			},
			FlushFunc: func(context.Context) error {
				if b.started > 0 {
					b.started--
				}
				return status(C.pipe_hip_flush(b.p), "flush")
			},
			ProcessFunc: func(signal.Floating, signal.Floating) (int, error) {
				return 0, errors.New("hip.Batch: run the Lines with pipe.RunBatched")
			},
			Batch: b, BatchSlot: slot, // the two fields ../pipe_patch adds to pipe.Processor
		}, nil
	}
}

// ProcessLines is the body of one stage-major pass: ins[slot] / outs[slot] are the pool buffers
// of the Lines that delivered a buffer this pass (nil: the Line has ended).  Every Line advances
// by exactly its own frames (a short read mid-stream included, pipe.go:404-406).
func (b *Batch) ProcessLines(ins, outs []signal.Floating) ([]int, error) {
	b.dirty = true
	inPtrs := make([]unsafe.Pointer, b.lines)
	outPtrs := make([]unsafe.Pointer, b.lines)
	for i := 0; i < b.lines; i++ {
		b.frames[i], b.live[i] = 0, ins[i] != nil
		if !b.live[i] {
			continue
		}
		b.frames[i] = C.int32_t(read(ins[i], b.in[i]))
		inPtrs[i], outPtrs[i] = b.inP[i], b.outP[i]
	}
	st := C.pipe_hip_process_lines_pinned(b.p, (*unsafe.Pointer)(unsafe.Pointer(&inPtrs[0])), &b.frames[0],
		(*unsafe.Pointer)(unsafe.Pointer(&outPtrs[0])), &b.written[0])
	runtime.KeepAlive(inPtrs)
	runtime.KeepAlive(outPtrs)
	if err := status(st, "process_lines"); err != nil {
		return nil, err
	}
	n := make([]int, b.lines)
	for i := 0; i < b.lines; i++ {
		if b.live[i] {
			n[i] = int(b.written[i])
			write(b.out[i][:n[i]*b.chans], outs[i])
		}
	}
	return n, nil
}

// Mutations address the shared handle: every Line of the group sees them at the same pass.
func (b *Batch) SetGain(g float64) mutable.Mutation {
	return b.mctx.Mutate(func() error {
		return status(C.pipe_hip_set_param(b.p, C.PIPE_HIP_PARAM_GAIN, (*C.double)(unsafe.Pointer(&g)), 1), "set gain")
	})
}
