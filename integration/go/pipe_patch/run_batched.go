package pipe

// Patch for the reference (package pipe, next to run.go): the stage-major sibling of
// multiLineExecutor.execute (run.go:112-132) that lets Processors sharing a device handle advance
// with ONE call per pass.  It is written against the reference's own unexported types (executor,
// lineExecutor, multiLineExecutor, Processor.in / .out, fitting.Message) and compiles inside that
// package once pipe.go carries the three-line change below.  SOURCE ONLY here: this image has no Go
// toolchain.  The compiled and tested equivalent is pipe::RunBatched / stageMajorExecutor in
// pipe_amd/csrc/host/pipe.cpp (tests/test_host_pipe.py: identical results to pipe.Run, Lines of
// different lengths, EOF removal, restart, live AddLine / InsertProcessor).
//
// The change to pipe.go (type Processor, pipe.go:49-60) -- two fields, nil / 0 for every existing
// Processor, so nothing else in the package changes behaviour:
//
//	Processor struct {
//		mutable.Context
//		ProcessFunc
//		StartFunc
//		FlushFunc
//		SignalProperties
//	+	Batch     BatchGroup // Processors with the same Batch advance with one ProcessLines call
//	+	BatchSlot int        // this Line's slot in it
//		in
//		out
//	}

import (
	"context"
	"io"

	"pipelined.dev/pipe/internal/fitting"
	"pipelined.dev/pipe/mutable"
	"pipelined.dev/signal"
)

// BatchGroup is implemented by hip.Batch (../hip/batched.go).  ins / outs have Slots() entries; a nil
// entry is a Line that brings no buffer this pass.  processed[slot] is what ProcessFunc would return.
type BatchGroup interface {
	Slots() int
	ProcessLines(ins, outs []signal.Floating) (processed []int, err error)
}

// pending is a batched Processor between the two halves of its execute (pipe.go:423-451).
type pending struct {
	proc   *Processor
	line   int // index into stageMajorExecutor.executors
	m      fitting.Message
	output signal.Floating
}

// batchBegin is pipe.go:424-437: receive, apply the in-band mutations, take an output buffer.
// io.EOF: the upstream fitting is closed; this stage's sender is closed in turn.
func (p *Processor) batchBegin(ctx context.Context) (pending, error) {
	m, ok := p.in.receiver.Receive(ctx)
	if !ok {
		p.out.sender.Close()
		return pending{}, io.EOF
	}
	if err := m.Mutations.ApplyTo(p.Context); err != nil {
		m.Signal.Free(p.in.allocator)
		return pending{}, err
	}
	return pending{proc: p, m: m, output: p.out.allocator.Float64()}, nil
}

// batchEnd is pipe.go:438-450 with ProcessFunc's results handed in: slice, send, free the input.
func (p *Processor) batchEnd(ctx context.Context, pd pending, processed int, procErr error) error {
	defer pd.m.Signal.Free(p.in.allocator) // the deferred Free of pipe.go:431
	if procErr != nil {
		p.out.sender.Close()
		pd.output.Free(p.out.allocator)
		return procErr
	}
	output := pd.output
	if processed != p.out.allocator.Length {
		output = output.Slice(0, processed)
	}
	if !p.out.sender.Send(ctx, fitting.Message{Signal: output, Mutations: pd.m.Mutations}) {
		p.out.sender.Close()
		output.Free(p.out.allocator)
		return io.EOF
	}
	return nil
}

// stageMajorExecutor runs the same Lines as multiLineExecutor with the same start / flush hooks and
// the same EOF -> flush -> remove rule; only the order inside one pass differs: every live Line's
// Source, then stage 1 of every Line, stage 2 ... then every Sink.  Lines share no state
// (run.go:112-132 runs them one after the other for no other reason than having one goroutine), so
// each Line sees exactly the data flow of pipe.Run.
type stageMajorExecutor struct {
	multiLineExecutor
	errs    []error                  // per Line, this pass: nil, io.EOF (closing) or the failure
	groups  map[BatchGroup][]pending // batched Processors of the current stage
	order   []BatchGroup             // ... in first-seen order (map iteration is random)
	ins     []signal.Floating
	outs    []signal.Floating
}

// RunBatched is pipe.Run (pipe.go:89-103) with the stage-major pass.
func RunBatched(ctx context.Context, bufferSize int, lines ...Line) error {
	e := stageMajorExecutor{groups: make(map[BatchGroup][]pending)}
	mctx := mutable.Mutable()
	for i, l := range lines {
		l.Context = mctx // one mutable context: sync fittings between the stages of every Line
		r, err := l.route(bufferSize)
		if err != nil {
			return err
		}
		r.connect(bufferSize)
		e.executors = append(e.executors, r.executor(nil, i))
	}
	return run(ctx, &e)
}

// note records the outcome of one component of Line i: the first real failure wins, io.EOF only
// marks the Line as closing (lineExecutor.execute keeps going on io.EOF so that the closure reaches
// the Sink, and returns a real error at once: run.go:37-52).
func (e *stageMajorExecutor) note(i int, err error) {
	if err == nil {
		return
	}
	if e.errs[i] == nil || e.errs[i] == io.EOF {
		e.errs[i] = err
	}
}

func (e *stageMajorExecutor) failed(i int) bool { return e.errs[i] != nil && e.errs[i] != io.EOF }

func (e *stageMajorExecutor) execute(ctx context.Context) error {
	n := len(e.executors)
	if cap(e.errs) < n {
		e.errs = make([]error, n)
	}
	e.errs = e.errs[:n]
	depth := 0
	for i, le := range e.executors {
		e.errs[i] = nil
		if le.started > depth {
			depth = le.started
		}
	}
	for stage := 0; stage < depth; stage++ {
		e.order = e.order[:0]
		for i, le := range e.executors {
			if stage >= le.started || e.failed(i) {
				continue // (a Line that failed runs no further component this pass: run.go:48)
			}
			p, batched := le.executors[stage].(*Processor)
			if !batched || p.Batch == nil {
				e.note(i, le.executors[stage].execute(ctx))
				continue
			}
			pd, err := p.batchBegin(ctx)
			if err != nil {
				e.note(i, err)
				continue
			}
			pd.line = i
			if _, seen := e.groups[p.Batch]; !seen {
				e.order = append(e.order, p.Batch)
			}
			e.groups[p.Batch] = append(e.groups[p.Batch], pd)
		}
		// one ProcessLines call -- one launch -- per group; then the second half of every execute.
		// A failure still runs batchEnd for every pending of the group: inputs go back to their
		// pools and the senders are closed.
		for _, g := range e.order {
			pds := e.groups[g]
			slots := g.Slots()
			if cap(e.ins) < slots {
				e.ins, e.outs = make([]signal.Floating, slots), make([]signal.Floating, slots)
			}
			ins, outs := e.ins[:slots], e.outs[:slots]
			for s := range ins {
				ins[s], outs[s] = nil, nil
			}
			for _, pd := range pds {
				ins[pd.proc.BatchSlot], outs[pd.proc.BatchSlot] = pd.m.Signal, pd.output
			}
			processed, err := g.ProcessLines(ins, outs)
			for _, pd := range pds {
				done := 0
				if err == nil {
					done = processed[pd.proc.BatchSlot]
				}
				e.note(pd.line, pd.proc.batchEnd(ctx, pd, done, err))
			}
			delete(e.groups, g)
		}
	}
	// retire the Lines exactly as multiLineExecutor.execute does (run.go:116-131): a Line that
	// returned io.EOF is flushed and removed; the first real failure ends the run
	for i := 0; i < len(e.executors); {
		err := e.errs[i]
		if err == nil {
			i++
			continue
		}
		if err != io.EOF {
			return err
		}
		if flushErr := e.executors[i].flushHook(ctx); flushErr != nil {
			return flushErr
		}
		e.executors = append(e.executors[:i], e.executors[i+1:]...)
		e.errs = append(e.errs[:i], e.errs[i+1:]...)
	}
	if len(e.executors) == 0 {
		return io.EOF
	}
	return nil
}

// Live edits keep their reference form: multiLineExecutor.addRoute / startSyncProcessor
// (run.go:134-169) are mutations applied between two passes, and the embedded multiLineExecutor
// provides them unchanged.  For a batched group AddLine claims a free slot of the group's handle:
// the allocator is hip.Batch.Allocator(slot); its StartFunc starts that slot alone when the group
// has already run a pass (pipe_hip_start_lines), so the Lines already streaming are not disturbed.
