package pipe

// PATCH SKETCH for the reference (package pipe): the stage-major sibling of
// multiLineExecutor.execute (run.go:112-132) that lets Processors sharing a device handle advance
// with ONE call per pass.  SOURCE ONLY, never compiled.  The compiled, tested equivalent is
// pipe::RunBatched / stageMajorExecutor in pipe_amd/csrc/host/pipe.cpp (tests/test_host_pipe.py:
// identical results to pipe.Run, incl. Lines of different lengths, EOF removal, restart).
//
// What it needs from the reference, and nothing else:
//   1. two fields on Processor:   Batch BatchGroup; BatchSlot int     (nil / -1 for ordinary ones)
//   2. Processor.execute (pipe.go:423-451) split around the ProcessFunc call:
//        batchBegin: Receive, ApplyTo(Context), allocate the output   (pipe.go:424-437)
//        batchEnd:   Slice to `processed`, Send, Free the input        (pipe.go:438-450)
//      execute() itself becomes batchBegin + ProcessFunc + batchEnd.

import (
	"context"
	"io"

	"pipelined.dev/signal"
)

// BatchGroup is implemented by hip.Batch.
type BatchGroup interface {
	Slots() int
	ProcessLines(ins, outs []signal.Floating) (processed []int, err error)
}

type pending struct {
	proc *Processor
	m    fittingMessage // the received message: Signal + Mutations
	out  signal.Floating
}

type stageMajorExecutor struct {
	multiLineExecutor // same Lines, same start / flush hooks, same EOF -> flush -> remove rule
}

// RunBatched is pipe.Run (pipe.go:89-103) with the stage-major pass.
func RunBatched(ctx context.Context, bufferSize int, lines ...Line) error {
	// bind exactly as Run does (one mutable context for all Lines => sync fittings), then
	// run(ctx, &stageMajorExecutor{...}) instead of run(ctx, &multiLineExecutor{...}).
	panic("sketch")
}

func (e *stageMajorExecutor) execute(ctx context.Context) error {
	// 1. every live Line's Source (run.go:113-119 per Line; io.EOF retires the Line:
	//    its remaining stages see the closed fitting, it is flushed and removed, run.go:120-128)
	// 2. for p := 0; p < deepest chain; p++:
	//      for every live Line with a stage p:
	//          ordinary Processor:  proc.execute(ctx)
	//          batched Processor:   proc.batchBegin(ctx) -> pending, collected per BatchGroup
	//      for every group:  processed, err := group.ProcessLines(ins, outs)   // ONE launch
	//                        for every pending of the group: proc.batchEnd(ctx, pending, processed[slot], err)
	//      (a failure still runs batchEnd for every pending of every group: inputs go back to
	//       their pools, senders are closed -- the deferred Free of pipe.go:431)
	// 3. every live Line's Sink
	// returns io.EOF when no Line is left (run.go:129-131)
	_ = io.EOF
	panic("sketch")
}

// Live edits keep their reference form: multiLineExecutor.addRoute / startSyncProcessor
// (run.go:134-169) are mutations applied between two passes.  For a batched group AddLine claims
// a free slot of the group's handle: the allocator is hip.Batch.Allocator(slot) and its StartFunc
// zeroes that slot's state only (pipe_hip_start_lines(handle, slot, 1)); the Lines already
// running are not disturbed.
