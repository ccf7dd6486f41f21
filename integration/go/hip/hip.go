// Package hip provides pipe.ProcessorAllocatorFuncs whose ProcessFunc bodies run on an AMD
// MI355X through libpipe_hip.so (C ABI: include/pipe_hip.h).
//
// SOURCE ONLY: the image this repository is built in has no Go toolchain and the reference's
// buffer package pipelined.dev/signal v0.10.0 is not vendored, so this file has never been
// compiled.  It is the binding a maintainer adds next to the reference; nothing in pipe.go,
// line.go, run.go, internal/fitting or mutable changes.  The same sequence of C-ABI calls is
// exercised from compiled code by the C++ host mirror (pipe_amd/csrc/host/hip_processors.cpp)
// and by examples/fir_stream.c.
//
// Every constructor returns a *Stage.  Stage.Allocator() is the pipe.ProcessorAllocatorFunc
// (line.go:26-30); once the pipe has bound the Line (pipe.New / pipe.Run call the allocator,
// line.go:62-104) the Stage holds the device handle and hands out mutations for it:
//
//	fir := hip.Fir(taps, hip.Options{Device: 0})
//	line := pipe.Line{Source: src, Processors: pipe.Processors(fir.Allocator()), Sink: sink}
//	p, _ := pipe.New(4096, line)
//	errc := p.Start(ctx)
//	p.Push(fir.SetTaps(newTaps))      // applied right before the ProcessFunc of the buffer
//	err := pipe.Wait(errc)            // it travels with (pipe.go:433)
package hip

/*
#cgo CFLAGS:  -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../pipe_amd/lib -lpipe_hip -Wl,-rpath,${SRCDIR}/../../../pipe_amd/lib
#include <stdlib.h>
#include "pipe_hip.h"
*/
import "C"

import (
	"context"
	"errors"
	"fmt"
	"runtime"
	"time"
	"unsafe"

	"pipelined.dev/pipe"
	"pipelined.dev/pipe/mutable"
	"pipelined.dev/signal"
)

// Options of every allocator.
type Options struct {
	Device int // HIP device ordinal; Line i of an n-GPU host takes i % n (SURVEY.md 8e)
	// Float32: the samples cross PCIe and sit in HBM as float32 (BASELINE.json's north star: "interleaved
	// multi-channel float32 buffers are DMA'd into HBM"; the pipe's own buffers stay signal.Floating = float64).
	// Arithmetic stays float64; the result is within 1 ulp OF FLOAT32 of the float64 chain (the bound of
	// PIPE_HIP_PARAM_EXACT in pipe_hip.h).  Half the bytes per call, and the forms that are float32-only apply:
	// a biquad buffer costs 27 us instead of 100 (the ordered float64 recurrence is one wave's issue).  Default
	// false: float64 buffers, bit for bit the oracle's float64 chain.
	Float32 bool
	// RelaxedFloat64: float64 buffers (the pipe's own: pipe.go:394,437) may take the forms that reassociate float64
	// arithmetic (PIPE_HIP_PARAM_RELAXED_F64): the biquad's tile form -- per 4096 x 2 call 22 us instead of 95, results
	// within 256 kappa 2^-53 of the Line's full scale of the ordered recurrence's instead of bit for bit -- and, for
	// batches (hip.Batch / BatchedChain), the FIR's overlap-save form -- results within 64 * 2^-53 * ||h||_1 * max|x| of
	// the ordered sum's, several times the ordered form's rate.  Off by default.
	RelaxedFloat64 bool
}

// Stage is one GPU Processor: its allocator before the Line is bound, its handle afterwards.
type Stage struct {
	create func(cfg *C.pipe_hip_config) (*C.pipe_hip_processor, error)
	opts   Options

	// set by the allocator (bind time)
	p    *C.pipe_hip_processor
	mctx mutable.Context // the context the pipe chose for this component (line.go:133-151
	// overwrite whatever the allocator returns, so mutations must be made with THIS one)
	inH, outH   []float64      // pinned staging, bufferSize * channels each (Options.Float32: inF / outF instead)
	inF, outF   []float32
	scratch     []float64      // Options.Float32: the float64 side of the conversion loops (bufferSize * max channels)
	inP, outP   unsafe.Pointer // their C addresses
	outChannels int
}

func status(st C.int, what string) error {
	if st == C.PIPE_HIP_OK {
		return nil
	}
	msg := C.GoString(C.pipe_hip_strerror(st))
	if st == C.PIPE_HIP_EHIP {
		return fmt.Errorf("pipe_hip %s: %s (hipError %d)", what, msg, int(C.pipe_hip_last_hip_error()))
	}
	return fmt.Errorf("pipe_hip %s: %s", what, msg)
}

func pinned(n int) ([]float64, unsafe.Pointer, error) {
	var p unsafe.Pointer
	if err := status(C.pipe_hip_host_alloc(C.int64_t(n*8), &p), "host_alloc"); err != nil {
		return nil, nil, err
	}
	return unsafe.Slice((*float64)(p), n), p, nil
}

func doubles(v []float64) *C.double {
	if len(v) == 0 {
		return nil
	}
	return (*C.double)(unsafe.Pointer(&v[0]))
}

func (o Options) dtype() int {
	if o.Float32 {
		return C.PIPE_HIP_F32
	}
	return C.PIPE_HIP_F64
}

// The pipe carries float64 buffers (pipe.go:394,437): dtype F64 unless Options.Float32, one Line per handle.
func (o Options) config(bufferSize, channels, lines int) C.pipe_hip_config {
	return C.pipe_hip_config{
		device: C.int32_t(o.Device), buffer_size: C.int32_t(bufferSize), channels: C.int32_t(channels),
		dtype: C.int32_t(o.dtype()), lines: C.int32_t(lines), max_batch: 1,
	}
}

// signal.Floating <-> the pinned staging slice, in BULK: signal.ReadFloat64(src Floating, dst []float64) and
// signal.WriteFloat64(src []float64, dst Floating) are what the reference's own tests move buffers with
// (mock/mock_test.go:120,128).  Rounds 1-4 of this shim copied one Sample(i) / SetSample(i, v) interface call at a
// time -- 16 384 calls per 4096 x 2 buffer, measured through the C++ stand-in at 26-28 us per buffer, MORE than the
// 20 us device round trip it wraps; the bulk copies are 3.7 us (INTEGRATION.md "What the binding costs").
func read(in signal.Floating, dst []float64) int {
	n := in.Length() * in.Channels()
	signal.ReadFloat64(in, dst[:n])
	return in.Length()
}

func write(src []float64, out signal.Floating) {
	signal.WriteFloat64(src, out)
}

// the same through float32 staging (Options.Float32): one bulk read into a float64 scratch slice, then plain loops
// over slices (no interface call per sample); the one rounding of the input happens here
func (s *Stage) read32(in signal.Floating, dst []float32) int {
	n := in.Length() * in.Channels()
	signal.ReadFloat64(in, s.scratch[:n])
	for i, v := range s.scratch[:n] {
		dst[i] = float32(v)
	}
	return in.Length()
}

func (s *Stage) write32(src []float32, out signal.Floating) {
	w := s.scratch[:len(src)]
	for i, v := range src {
		w[i] = float64(v)
	}
	signal.WriteFloat64(w, out)
}

func pinned32(n int) ([]float32, unsafe.Pointer, error) {
	var p unsafe.Pointer
	if err := status(C.pipe_hip_host_alloc(C.int64_t(n*4), &p), "host_alloc"); err != nil {
		return nil, nil, err
	}
	return unsafe.Slice((*float32)(p), n), p, nil
}

// Allocator is the pipe.ProcessorAllocatorFunc of this stage (line.go:26-30).  It may run while
// the pipe is running (pipe.go:314-321): everything it touches belongs to this Stage.
func (s *Stage) Allocator() pipe.ProcessorAllocatorFunc {
	return func(mctx mutable.Context, bufferSize int, in pipe.SignalProperties) (pipe.Processor, error) {
		cfg := s.opts.config(bufferSize, in.Channels, 1)
		p, err := s.create(&cfg)
		if err != nil {
			return pipe.Processor{}, err // pipe.New wraps it: "processor: %w" (line.go:72-74)
		}
		var ch, up, down C.int32_t
		if err := status(C.pipe_hip_output_properties(p, &ch, &up, &down), "output_properties"); err != nil {
			C.pipe_hip_destroy(p)
			return pipe.Processor{}, err
		}
		// the finalizer goes on BEFORE anything else can fail: a failed pinned allocation below
		// must not leak the device handle (Close frees whatever exists, nil pointers included)
		s.p, s.mctx, s.outChannels = p, mctx, int(ch)
		runtime.SetFinalizer(s, (*Stage).Close) // Go has no destructor hook on a Processor
		if s.opts.RelaxedFloat64 && !s.opts.Float32 {
			one := C.double(1)
			// (a stage with neither a FIR nor a biquad in it answers PIPE_HIP_EINVAL: nothing to relax)
			C.pipe_hip_set_param(p, C.PIPE_HIP_PARAM_RELAXED_F64, &one, 1)
		}
		if s.opts.Float32 {
			if s.inF, s.inP, err = pinned32(bufferSize * in.Channels); err == nil {
				s.outF, s.outP, err = pinned32(bufferSize * int(ch))
			}
			mc := in.Channels
			if int(ch) > mc {
				mc = int(ch)
			}
			s.scratch = make([]float64, bufferSize*mc)
		} else {
			if s.inH, s.inP, err = pinned(bufferSize * in.Channels); err == nil {
				s.outH, s.outP, err = pinned(bufferSize * int(ch))
			}
		}
		if err != nil {
			s.Close()
			return pipe.Processor{}, err
		}
		return pipe.Processor{
			// the stage's OUTPUT properties: they size the out pool (pipe.go:418) and are the
			// next stage's input (line.go:75)
			SignalProperties: pipe.SignalProperties{
				SampleRate: in.SampleRate * signal.Frequency(up) / signal.Frequency(down),
				Channels:   int(ch),
			},
			// run.go:64-74: once, inside the executing goroutine, before the loop; a pipe may be
			// started again (pipe_test.go:108-131) and pipe_hip_start zeroes the per-Line state.
			// Every C entry point selects its device itself: goroutines migrate between OS
			// threads and HIP's current device is per thread, no LockOSThread needed.
			StartFunc: func(context.Context) error { return status(C.pipe_hip_start(s.p), "start") },
			// pipe.go:438: one buffer.  `in` is freed by the pipe right after the call
			// (pipe.go:431) and must not be kept; `out` has Length() == bufferSize.
			ProcessFunc: func(in, out signal.Floating) (int, error) {
				var n int
				if s.opts.Float32 {
					n = s.read32(in, s.inF)
				} else {
					n = read(in, s.inH)
				}
				var written C.int32_t
				st := C.pipe_hip_process(s.p, s.inP, C.int32_t(n), s.outP, C.int32_t(out.Length()), &written)
				if err := status(st, "process"); err != nil {
					return 0, err // the run ends with "error running: %w" (run.go:191-193)
				}
				if s.opts.Float32 {
					s.write32(s.outF[:int(written)*s.outChannels], out)
				} else {
					write(s.outH[:int(written)*s.outChannels], out)
				}
				return int(written), nil // pipe.go:441-443 slices out when written < bufferSize
			},
			FlushFunc: func(context.Context) error { return status(C.pipe_hip_flush(s.p), "flush") },
		}, nil
	}
}

// Close releases device and pinned memory (also the finalizer).
func (s *Stage) Close() {
	// (the allocator may run again on the same Stage -- a pipe rebuilt after an error: SetFinalizer on an object
	// that still has one is fatal in Go, so Close clears it)
	runtime.SetFinalizer(s, nil)
	if s.p != nil {
		C.pipe_hip_destroy(s.p)
		s.p = nil
	}
	// (pipe_hip_host_free accepts NULL: an allocator that failed half-way leaves one of them nil)
	C.pipe_hip_host_free(s.inP)
	C.pipe_hip_host_free(s.outP)
	s.inP, s.outP, s.inH, s.outH, s.inF, s.outF = nil, nil, nil, nil, nil, nil
}

// ---- allocators -----------------------------------------------------------------------------

// Gain: y = x * gain.  Gain(1, o) is the reference's mock.Processor (mock/mock.go:139-157).
func Gain(gain float64, o Options) *Stage {
	return &Stage{opts: o, create: func(cfg *C.pipe_hip_config) (*C.pipe_hip_processor, error) {
		var p *C.pipe_hip_processor
		return p, status(C.pipe_hip_gain_create(cfg, C.double(gain), &p), "gain_create")
	}}
}

// Fir: direct-form FIR, the same taps for every channel.
func Fir(taps []float64, o Options) *Stage {
	taps = append([]float64(nil), taps...)
	return &Stage{opts: o, create: func(cfg *C.pipe_hip_config) (*C.pipe_hip_processor, error) {
		var p *C.pipe_hip_processor
		st := C.pipe_hip_fir_create(cfg, doubles(taps), C.int32_t(len(taps)), &p)
		runtime.KeepAlive(taps)
		return p, status(st, "fir_create")
	}}
}

// Biquad: DF2T cascade; coeffs holds {b0, b1, b2, a1, a2} per section.
func Biquad(coeffs []float64, o Options) *Stage {
	coeffs = append([]float64(nil), coeffs...)
	return &Stage{opts: o, create: func(cfg *C.pipe_hip_config) (*C.pipe_hip_processor, error) {
		if len(coeffs)%5 != 0 {
			return nil, errors.New("hip.Biquad: coefficients come in fives")
		}
		var p *C.pipe_hip_processor
		st := C.pipe_hip_biquad_create(cfg, doubles(coeffs), C.int32_t(len(coeffs)/5), &p)
		runtime.KeepAlive(coeffs)
		return p, status(st, "biquad_create")
	}}
}

// Resampler: rational polyphase, output SampleRate = input * up / down (line.go:38-41 carries
// it to the next stage and to the sink).  proto has up*tapsPerPhase taps.  An up-sampler emits
// more frames than it reads, which ProcessFunc cannot express with full buffers (out is
// bufferSize frames, pipe.go:437-443): feed it at most floor(bufferSize*down/up) frames per
// buffer (3763 for 4096 at 44.1 -> 48 kHz) or the call fails with "output exceeds buffer capacity".
func Resampler(proto []float64, tapsPerPhase, up, down int, o Options) *Stage {
	proto = append([]float64(nil), proto...)
	return &Stage{opts: o, create: func(cfg *C.pipe_hip_config) (*C.pipe_hip_processor, error) {
		var p *C.pipe_hip_processor
		st := C.pipe_hip_resampler_create(cfg, doubles(proto), C.int32_t(tapsPerPhase), C.int32_t(up),
			C.int32_t(down), &p)
		runtime.KeepAlive(proto)
		return p, status(st, "resampler_create")
	}}
}

// FeedFor adapts a Source to an up-sampling Resampler further down its Line.  The pipe hands every component
// buffers of the Line's bufferSize frames (pipe.go:90,107,437-443), so a stage that emits up/down times the
// frames it reads fits its output only if it reads at most floor(bufferSize*down/up) of them: FeedFor allocates
// the wrapped Source for that many frames and lets it fill only the head of the pipe's buffer -- a short read,
// which the pipe carries on with (pipe.go:404-406: Slice(0, read)); every Processor behind it then sees
// buffers the Resampler's output fits (3763 of 4096 frames at 44.1 -> 48 kHz).  For up <= down it returns the
// Source unchanged.
//
//	pipe.Line{Source: hip.FeedFor(src, 160, 147), Processors: []pipe.ProcessorAllocatorFunc{rs.Allocator()}, Sink: sink}
func FeedFor(src pipe.SourceAllocatorFunc, up, down int) pipe.SourceAllocatorFunc {
	if up <= down {
		return src
	}
	return func(mctx mutable.Context, bufferSize int) (pipe.Source, error) {
		feed := bufferSize * down / up
		if feed < 1 {
			return pipe.Source{}, fmt.Errorf("hip.FeedFor: bufferSize %d too small for %d/%d", bufferSize, up, down)
		}
		s, err := src(mctx, feed)
		if err != nil {
			return pipe.Source{}, err
		}
		inner := s.SourceFunc
		s.SourceFunc = func(out signal.Floating) (int, error) {
			// the wrapped Source fills at most `feed` frames of the pipe's buffer
			return inner(out.Slice(0, feed))
		}
		return s, nil
	}
}

// Chain: several fixed-rate stages (Fir, Biquad, Gain) as ONE Processor whose intermediates stay
// on the device (a Line's Processors slice, line.go:17, collapsed into one component).  The
// stages' own Allocators must not be used as well.
func Chain(o Options, stages ...*Stage) *Stage {
	return &Stage{opts: o, create: func(cfg *C.pipe_hip_config) (*C.pipe_hip_processor, error) {
		raw := make([]*C.pipe_hip_processor, 0, len(stages))
		destroy := func() {
			for _, r := range raw {
				C.pipe_hip_destroy(r)
			}
		}
		for _, st := range stages {
			p, err := st.create(cfg)
			if err != nil {
				destroy()
				return nil, err
			}
			raw = append(raw, p)
		}
		var chain *C.pipe_hip_processor
		st := C.pipe_hip_chain_create(&raw[0], C.int32_t(len(raw)), &chain) // takes ownership
		runtime.KeepAlive(raw)
		if err := status(st, "chain_create"); err != nil {
			destroy()
			return nil, err
		}
		return chain, nil
	}}
}

// ---- mutations (mutable/mutable.go:40-48; applied at pipe.go:433, before ProcessFunc) -------------

func (s *Stage) setParam(param C.int32_t, values []float64, what string) mutable.Mutation {
	values = append([]float64(nil), values...)
	return s.mctx.Mutate(func() error {
		st := C.pipe_hip_set_param(s.p, param, doubles(values), C.int32_t(len(values)))
		runtime.KeepAlive(values)
		return status(st, what)
	})
}

// SetGain / SetTaps / SetCoeffs: take effect for the buffer the mutation travels with and for
// none already processed.  The upload is asynchronous on the handle's own stream: other Lines on
// the same device are not stalled.  On a Chain the parameter goes to the first stage that takes it;
// SetStageParam addresses one stage.
func (s *Stage) SetGain(g float64) mutable.Mutation {
	return s.setParam(C.PIPE_HIP_PARAM_GAIN, []float64{g}, "set gain")
}
func (s *Stage) SetTaps(taps []float64) mutable.Mutation {
	return s.setParam(C.PIPE_HIP_PARAM_TAPS, taps, "set taps")
}
func (s *Stage) SetCoeffs(c []float64) mutable.Mutation {
	return s.setParam(C.PIPE_HIP_PARAM_COEFFS, c, "set coeffs")
}

// SetExact pins the ordered-fma (bit-exact) forms even for float32 batches.
func (s *Stage) SetExact(on bool) mutable.Mutation {
	v := 0.0
	if on {
		v = 1
	}
	return s.setParam(C.PIPE_HIP_PARAM_EXACT, []float64{v}, "set exact")
}

// SetResident asks for the device's DOORBELL for this stage: the NEXT buffer's work is then kept queued on the
// device ahead of its ProcessFunc call (PIPE_HIP_PARAM_RESIDENT: behind a doorbell word in pinned host memory; the
// call costs no kernel launch and no completion event -- gain 12.2 -> 9.3 us, a 256-tap FIR on 4096 x 2
// 19.5 -> 15.7 us).  ONE stage per device can hold it (a parked queue costs every other parked queue of the process
// tens of microseconds and holds up whatever shares its hardware queue: DESIGN.md section 5): give it to the stage
// of the pipe that is called most, or fuse a Line's stages with hip.Chain and give it to the chain.  A stage that
// asks while another one holds it stays on the plain path and the mutation reports ErrDoorbellBusy (not a failure of
// the stream: errors.Is it and go on); stages that cannot take a queued launch back (the resampler, long biquad
// cascades) report an error as well.  idle == 0: the library's 250 ms; work queued for longer than `idle` without a
// call is dropped by the library (a queue waiting for its doorbell holds up device-wide waits of the process);
// ResidentInfo counts such drops.  Results are bit for bit those of the plain path.
func (s *Stage) SetResident(on bool, idle time.Duration) mutable.Mutation {
	v := 0.0
	if on {
		v = 1
		if ms := float64(idle / time.Millisecond); ms > 1 {
			v = ms
		}
	}
	return s.mctx.Mutate(func() error {
		d := C.double(v)
		st := C.pipe_hip_set_param(s.p, C.PIPE_HIP_PARAM_RESIDENT, &d, 1)
		if st == C.PIPE_HIP_EBUSY {
			return ErrDoorbellBusy
		}
		return status(st, "set resident")
	})
}

// SetResidentShared puts the stage into its device's SHARED doorbell queue (PIPE_HIP_PARAM_RESIDENT_SHARED, round 6): up to
// sixteen stages of a device each keep their next buffer's work queued at the tail of ONE hardware queue, in the order
// they are called.  That order is the order of the last round exactly when the pipe runs its Lines synchronously --
// pipe.Run: all Lines of a mutable context in one goroutine, round-robin, a Line's stages in order (run.go:37-52,
// 112-132) -- and then EVERY stage of EVERY Line has the doorbell's latency (a 256-tap FIR call 20 us through a binding
// where the plain path takes 24 - 30 with 2 - 16 stages on the device).  In the asynchronous mode (pipe.New + Start: a
// goroutine per component) calls arrive in any order, ring each other's queued work and cost MORE than the plain path
// (27 against 12 us in the probe): use SetResident on the one busiest stage there.  ErrDoorbellBusy when a stage holds
// the device's doorbell exclusively or sixteen share it already.  Bit for bit the plain path's results.
func (s *Stage) SetResidentShared(on bool, idle time.Duration) mutable.Mutation {
	v := 0.0
	if on {
		v = 1
		if ms := float64(idle / time.Millisecond); ms > 1 {
			v = ms
		}
	}
	return s.mctx.Mutate(func() error {
		d := C.double(v)
		st := C.pipe_hip_set_param(s.p, C.PIPE_HIP_PARAM_RESIDENT_SHARED, &d, 1)
		if st == C.PIPE_HIP_EBUSY {
			return ErrDoorbellBusy
		}
		return status(st, "set resident shared")
	})
}

// ErrDoorbellBusy: another stage of the device holds the doorbell; this one keeps the plain path.
var ErrDoorbellBusy = errors.New("pipe_hip: the device's doorbell is held by another stage")

// ResidentInfo: does the stage hold its device's doorbell, and how many queued launches were run on stale input
// and dropped -- by the idle watchdog / by another entry (a mutation, a short buffer, Start, Flush).  A host that
// sees the first count grow feeds the stage slower than the idle limit.
func (s *Stage) ResidentInfo() (holds bool, droppedByWatchdog, droppedByEntry int64) {
	var h C.int32_t
	var w, e C.int64_t
	C.pipe_hip_resident_info(s.p, &h, &w, &e)
	return h != 0, int64(w), int64(e)
}

// SetStageParam: parameter `param` (C.PIPE_HIP_PARAM_*) of stage `stage` of a Chain.
func (s *Stage) SetStageParam(stage int, param int, values []float64) mutable.Mutation {
	values = append([]float64(nil), values...)
	return s.mctx.Mutate(func() error {
		st := C.pipe_hip_chain_set_param(s.p, C.int32_t(stage), C.int32_t(param), doubles(values),
			C.int32_t(len(values)))
		runtime.KeepAlive(values)
		return status(st, "chain set param")
	})
}

// ---- n-input mix -----------------------------------------------------------------------------
// The reference has no signal merger (merger.go merges error channels), and ProcessFunc has one
// input, so the n-input sum is not a Processor.  Mix is a helper for a custom component that owns
// n upstream buffers of equal length.
type Mix struct {
	p *C.pipe_hip_processor
}

func NewMix(inputs, bufferSize, channels int, o Options) (*Mix, error) {
	cfg := o.config(bufferSize, channels, 1)
	var p *C.pipe_hip_processor
	if err := status(C.pipe_hip_mix_create(&cfg, C.int32_t(inputs), &p), "mix_create"); err != nil {
		return nil, err
	}
	return &Mix{p: p}, nil
}

// Sum writes ((ins[0] + ins[1]) + ins[2]) ... into out; all slices hold frames*channels samples.
func (m *Mix) Sum(ins [][]float64, frames int, out []float64) error {
	ptrs := make([]unsafe.Pointer, len(ins))
	pin := runtime.Pinner{}
	defer pin.Unpin()
	for i := range ins {
		pin.Pin(&ins[i][0])
		ptrs[i] = unsafe.Pointer(&ins[i][0])
	}
	st := C.pipe_hip_mix_process(m.p, (*unsafe.Pointer)(unsafe.Pointer(&ptrs[0])), C.int32_t(len(ins)),
		C.int32_t(frames), unsafe.Pointer(&out[0]))
	runtime.KeepAlive(ins)
	runtime.KeepAlive(out)
	return status(st, "mix_process")
}

func (m *Mix) Close() { C.pipe_hip_destroy(m.p) }
