"""Training engine: one process per GPU, data-parallel over utterances with RCCL.

The reference trains on a single device through ``tf.estimator`` (``asr/train.py:31-55``,
``train_distribute=None``); the step it runs is forward -> CTC loss -> backward -> Adam
(``asr/model.py:49-83``).  `Trainer.train_step` is that step written out over the HIP kernels,
plus what BASELINE.json adds: minibatches sharded across the GPUs of a node, gradients summed
with RCCL all-reduce over xGMI — per layer slice of the flat gradient arena, launched as soon
as the backward pass has finished that layer so the collective overlaps the rest of backward.
"""

import collections
import warnings
import os
import time

import torch
import torch.distributed as dist

from ctc_asr_amd.model import CTCModel


def init_distributed(backend=None):
    """Initialise ``torch.distributed`` from the torchrun environment (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*).  Returns (rank, local_rank, world_size).  ``backend`` defaults to
    'nccl' (= RCCL on ROCm) when a GPU is visible, else 'gloo'."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = os.environ.get('CTCASR_DIST_BACKEND') or \
                ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class NanLossDuringTrainingError(RuntimeError, FloatingPointError):
    """The training loss is NaN or infinite: what the reference's ``NanTensorHook`` raises
    (``asr/model.py:368``; ``tf.train.NanLossDuringTrainingError`` is a RuntimeError).  Raised by
    the deferred checks of `Trainer` (the guard word of the step that hit it, at most
    `max_steps_ahead` steps later) and by `train.train_epoch` where it reads the loss."""


class GradientReducer:
    """Bucketed sum-all-reduce of the flat gradient arena.

    ``hook(layer, start, stop)`` is called by ``CTCModel.backward`` when a layer's slice is
    final; slices are merged into buckets of at least ``bucket_bytes`` (contiguous because the
    arena is laid out in layer order and backward walks it from the end) and each bucket is
    reduced asynchronously.  ``finish()`` flushes the remainder and waits.  With world size 1
    everything is a no-op.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets of
    tens of MB keep each ring step bandwidth- rather than latency-bound without delaying the
    first launch."""

    def __init__(self, grad_arena, world_size, bucket_bytes=64 << 20, group=None,
                 hold_until=None, force=False, max_bucket_bytes=None, stand_in=None):
        self.grad, self.world, self.group = grad_arena, world_size, group
        # ``stand_in`` = (workgroups, GB/s): instead of a collective every bucket launches a
        # kernel of that many resident workgroups that holds its CUs for bytes / rate on a
        # stream of its own - ordered after the launching stream and waited for by `finish()`
        # exactly like an asynchronous all-reduce.  The one-GPU stand-in for RCCL's ring kernels
        # beside the persistent recurrences (NCCL refuses two ranks on one device); implies
        # `force`.  Gradients are left alone (a one-rank sum).
        self.stand_in = stand_in
        self._stand_in_stream = None
        self._stand_in_scratch = None
        # `force`: run the collectives even with a single rank (a world-size-1 process group is
        # legal): tests use it to put the real backend (RCCL) under the real backward pass on a
        # one-GPU box
        self.active = world_size > 1 or force or stand_in is not None
        self.bucket_elems = max(1, bucket_bytes // 4)
        # slices are merged until a bucket holds at least `bucket_bytes`; a pending range larger
        # than `max_bucket_bytes` (a held-back release: everything at once; or one big layer) is
        # cut into equal pieces no larger than that, so that the first piece's ring steps start
        # while the later ones are still being enqueued and no single collective is hundreds of MB
        self.max_bucket_elems = max(self.bucket_elems,
                                    (max_bucket_bytes or max(bucket_bytes, 64 << 20)) // 4)
        self.pending_start = None
        self.pending_stop = None
        self.works = []
        self.launched = 0               # all-reduce launches so far (bench.py reports per step)
        # Collective kernels must not be in flight while a persistent recurrence kernel starts:
        # that kernel needs all of its workgroups co-resident (one per CU) and spins at grid
        # barriers, so RCCL workgroups holding CUs could stall it past its spin limit.  With
        # `hold_until` = name of the last recurrent layer of the backward pass, buckets are only
        # launched from that layer's hook on - they then overlap the remaining weight-gradient
        # GEMMs and the conv backward, which have no residency requirement.
        self.hold_until = hold_until
        self.released = hold_until is None

    def __call__(self, layer, start, stop):
        self.hook(layer, start, stop)

    def flush(self):
        """Launch whatever is pending now (if released).  `CTCModel.backward` calls this at the
        end of the block of hooks that run with the side stream current: those slices were
        produced by side-stream GEMMs, so their all-reduce must be enqueued while that stream is
        still the current one - never merged into a bucket that is launched later from the main
        stream, which is not ordered after them."""
        if self.active and self.released:
            self._flush()

    def hook(self, layer, start, stop):
        if not self.active:
            return
        if layer == self.hold_until:
            self.released = True
        if self.pending_stop is None:
            self.pending_start, self.pending_stop = start, stop
        else:
            # backward visits layers from the end of the arena towards its start
            if stop != self.pending_start:       # not adjacent: cannot merge into one view
                self._flush()
                self.pending_start, self.pending_stop = start, stop
            else:
                self.pending_start = start
        if self.released and self.pending_stop - self.pending_start >= self.bucket_elems:
            self._flush()

    def _flush(self):
        if self.pending_stop is None:
            return
        start, stop = self.pending_start, self.pending_stop
        pieces = -(-(stop - start) // self.max_bucket_elems)
        piece = -(-(stop - start) // pieces)
        piece = (piece + 3) // 4 * 4                 # keep 16-byte alignment of every view
        hi = stop
        while hi > start:                            # from the end: the order backward made them
            lo = max(start, hi - piece)
            if self.stand_in is not None:
                self.works.append(self._launch_stand_in((hi - lo) * 4))
            else:
                self.works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM,
                                                  group=self.group, async_op=True))
            self.launched += 1
            hi = lo
        self.pending_start = self.pending_stop = None

    def _launch_stand_in(self, nbytes):
        """The stream semantics of `dist.all_reduce(async_op=True)` on the nccl backend: the
        collective runs on the backend's own stream behind everything enqueued so far on the
        launching stream; `wait()` makes the then-current stream wait for it."""
        from ctc_asr_amd import hip
        workgroups, gb_per_s = self.stand_in[:2]
        traffic = len(self.stand_in) > 2 and bool(self.stand_in[2])
        device = self.grad.device
        if traffic and self._stand_in_scratch is None:
            # (zeros: a += b stays zero however often it runs)
            self._stand_in_scratch = tuple(torch.zeros(16 << 20, dtype=torch.float32,
                                                       device=device) for _ in range(2))
        if self._stand_in_stream is None:
            self._stand_in_stream = torch.cuda.Stream(device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(device))
        with torch.cuda.stream(self._stand_in_stream):
            self._stand_in_stream.wait_event(ready)
            busy_us = max(1, int(nbytes / (gb_per_s * 1e3)))
            if traffic:
                hip.collective_traffic(self._stand_in_scratch[0], self._stand_in_scratch[1],
                                       workgroups, busy_us, nbytes)
            else:
                hip.occupy_cus(workgroups, busy_us)
            done = torch.cuda.Event()
            done.record(self._stand_in_stream)

        class _Work:
            @staticmethod
            def wait():
                torch.cuda.current_stream(device).wait_event(done)
        return _Work

    def finish(self, guard=None):
        """Flush and wait.  ``guard`` (the int32 words of `CTCModel.step_guard`): with more than
        one rank its MAXIMUM over ranks comes back in place - a step that one replica must not
        apply carries that replica's gradients in every rank's sum, so every replica drops it
        (the Adam kernel's skip flag) and the parameters stay replicated until the error the
        bad rank raises stops the job."""
        if not self.active:
            return
        self.released = self.hold_until is None
        self._flush()
        if guard is not None and self.world > 1 and self.stand_in is None:
            self.works.append(dist.all_reduce(guard, op=dist.ReduceOp.MAX, group=self.group,
                                              async_op=True))
        for work in self.works:
            work.wait()
        self.works = []


class Trainer:
    """Owns a `CTCModel` replica on this rank's GPU and runs synchronous data-parallel steps."""

    def __init__(self, cfg, flags=None, device=None, seed=0, params=None, world_size=1, rank=0,
                 bucket_bytes=64 << 20, conv_autotune=None, allreduce_early=None,
                 force_reducer=False, reduce=True, collective_stand_in=None):
        self.world, self.rank = world_size, rank
        device = device or 'cuda:{}'.format(torch.cuda.current_device())
        self.model = CTCModel(cfg, device, seed=seed, params=params, conv_autotune=conv_autotune)
        self.lr = getattr(flags, 'learning_rate', 1e-5) if flags is not None else 1e-5
        self.beta1 = getattr(flags, 'adam_beta1', 0.9) if flags is not None else 0.9
        self.beta2 = getattr(flags, 'adam_beta2', 0.999) if flags is not None else 0.999
        self.eps = getattr(flags, 'adam_epsilon', 1e-8) if flags is not None else 1e-8
        # Release of the gradient buckets (``allreduce_early``; None = CTCASR_ALLREDUCE_EARLY):
        #   early (default since round 6)
        #                   every layer's bucket is reduced as soon as its weight-gradient GEMMs
        #                   are done on the side stream, beside the recurrences of the layers
        #                   below.  RCCL's workgroups occupy CUs for as long as a collective runs;
        #                   the H = 1024 backward recurrence needs 128 of the 256 CUs, so both
        #                   fit - a persistent workgroup that finds its CU taken waits (spin
        #                   limit: seconds).
        #   held            nothing is reduced before the last persistent recurrence launch of
        #                   the backward pass has been issued (`hold_until='rnn0'`).
        # Why early: with a stand-in that has a ring all-reduce's CU footprint AND its memory
        # traffic beside the real step (one GPU) the C3 step pays +2.3 ms
        # (4.6 %) for early against +5.3 ms (10.4 %) for held, and neither mode timed out; held
        # leaves the whole 0.49 GB of gradients to be reduced behind the backward pass.  bench.py
        # measures both modes at N > 1 and reports the better one, so a node on which RCCL beside
        # the recurrences behaves worse than the stand-in still shows in the line.
        # Models whose recurrence takes ALL 256 CUs (LSTM / GRU at H = 2048) always hold: there
        # the hooks run on the main stream right in front of whole-chip persistent launches,
        # whose resident workgroups would spin until the collective has drained.
        if allreduce_early is None:
            allreduce_early = os.environ.get('CTCASR_ALLREDUCE_EARLY', '1') == '1'
        whole_chip = cfg.cell in ('lstm', 'gru') and cfg.num_units_rnn == 2048
        early = bool(allreduce_early) and not whole_chip
        active = (world_size > 1 or force_reducer or collective_stand_in is not None) and reduce
        self.model.early_hooks = early and active
        # ``reduce=False`` (bench.py's "stubbed" leg): the same step without any collective, to
        # price what the all-reduce adds to it; replicas drift apart - timing only
        self.reducer = GradientReducer(self.model.arena.grad, world_size if reduce else 1,
                                       bucket_bytes, hold_until=None if early else 'rnn0',
                                       force=force_reducer and reduce,
                                       stand_in=collective_stand_in if reduce else None)
        self.release = 'early' if early else 'held'
        if world_size > 1:   # identical replicas: rank 0's initial parameters win
            dist.broadcast(self.model.arena.param, src=0)
            self.model.arena.touch()
        # The host enqueues a step several times faster than the GPU runs it.  Left alone it
        # runs ahead without bound: every step's activations (5 GB at C3) are then allocated
        # anew - blocks freed by the host are still in use by steps the GPU has not reached -
        # until the allocator hits the end of HBM and synchronises.  `max_steps_ahead` bounds
        # the lead: step N is not enqueued before step N - max_steps_ahead has finished.
        self.max_steps_ahead = 2
        self._step_done = collections.deque()
        self.host_wait_s = 0.0          # time spent waiting there (bench.py reports the rest)
        # Deferred error checks (``train_step(check=True)``): a step's CTC status words travel to
        # pinned host memory with an asynchronous copy and are looked at when that step is known
        # to have finished - at the wait above, `max_steps_ahead` steps later - and the sticky
        # time-out word of the persistent recurrence kernels (no launch clears it) is polled
        # every `rnn_poll_every` steps and by `drain_checks()`.  Neither stalls the host, so the
        # run-ahead above also holds for real training, not only for the benchmark.
        self.rnn_poll_every = 50
        self._pending_status = collections.deque()
        self._skipped = None
        self._skipped_reconciled = 0
        self._steps_since_poll = 0

    def _check_finished_steps(self, wait=False):
        """Raise for any finished step whose CTC status reported an infeasible alignment /
        bad labels (``wait``: block until every pending step has finished)."""
        while self._pending_status:
            event, host, step = self._pending_status[0]
            if not wait and not event.query():
                break
            event.synchronize()
            self._pending_status.popleft()
            words = host.numpy()
            try:
                CTCModel.check_status(host[:-2])
            except ValueError as err:
                raise type(err)('{} [training step {}; its update was not applied]'.format(
                    err, step)) from None
            if words[-1] != 0:
                # a persistent recurrence gave up at a grid barrier in that step: the sticky word
                # is read (and the barrier words reset) by the synchronous poll, which raises
                self.model.check_rnn_error()
            if words[-2] != 0:
                raise NanLossDuringTrainingError(
                    'NaN loss during training. [non-finite CTC loss in training step {}; its '
                    'update was not applied]'.format(step))

    def drain_checks(self):
        """Wait for the steps in flight and raise what they have to report: CTC status of every
        pending step, then the persistent kernels' time-out word.  `train.train_epoch` calls
        this where it reads the loss anyway, at the end of an epoch and before a checkpoint."""
        self._check_finished_steps(wait=True)
        self.model.check_rnn_error()
        self._steps_since_poll = 0
        # Steps the device dropped without a check having raised for them (`train_step(check=
        # False)`, --no-step-checks): TensorFlow's global step would not have advanced - take them
        # back out of the step counter here, where the host is synchronised anyway, so that Adam's
        # bias correction and the step a checkpoint is written under count applied updates only
        # (ADVICE r05), and say so.
        dropped = self.skipped_step_count()
        if dropped > self._skipped_reconciled:
            new = dropped - self._skipped_reconciled
            self._skipped_reconciled = dropped
            self.model.step_count -= new
            warnings.warn('{} training step(s) were dropped on the device (CTC status, non-finite '
                          'loss or a kernel time-out); the step counter is back at {}.'.format(
                              new, self.model.step_count), RuntimeWarning)

    def train_step(self, features, feature_len, labels, check=True):
        """forward + CTC + backward (+ all-reduce) + Adam on this rank's shard of the global
        batch; returns the local mean loss (device scalar).  The global loss is the mean over
        ranks of the local means (equal shard sizes), so gradients are summed and scaled by
        1 / world_size inside the Adam kernel.

        ``check=True``: errors are raised without stalling the host - an infeasible alignment
        (where ``tf.nn.ctc_loss`` raises), a non-finite loss or a timed-out persistent kernel
        surfaces at most `max_steps_ahead` steps after the step that hit it (the guard word
        travels to pinned memory with the CTC status), and always before a checkpoint is
        written.  Either way that step's update is dropped ON THE DEVICE (`CTCModel.step_guard`
        -> the Adam kernel's skip flag): parameters and moments never see its gradients."""
        if len(self._step_done) >= self.max_steps_ahead:
            t0 = time.perf_counter()
            while len(self._step_done) >= self.max_steps_ahead:
                self._step_done.popleft().synchronize()
            self.host_wait_s += time.perf_counter() - t0
        if check:
            self._check_finished_steps()
            self._steps_since_poll += 1
            if self._steps_since_poll >= self.rnn_poll_every:
                # (synchronises: once per `rnn_poll_every` steps)
                self._check_finished_steps(wait=True)
                self.model.check_rnn_error()
                self._steps_since_poll = 0
        loss = self.model.forward_backward(features, feature_len, labels,
                                           reduce_hook=self.reducer, check=False)
        # Invalid gradients never reach the parameters: the guard word is computed on the device
        # from this step's CTC status, its loss and the recurrence time-out words, and the Adam
        # kernel drops the update when it is set - whenever the host gets to look.
        # (N > 1: the guard words are max-reduced with the gradients, every replica drops the
        # step that one of them must not apply; the rank that hit it raises.)
        guard = self.model.step_guard()
        if check:
            status = self.model.last_status
            host = torch.empty(status.numel() + 2, dtype=status.dtype, pin_memory=True)
            host[:status.numel()].copy_(status, non_blocking=True)
            host[status.numel():].copy_(guard, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(torch.cuda.current_stream(self.model.device))
            self._pending_status.append((copied, host, self.model.step_count + 1))
        self.reducer.finish(guard)      # (N > 1: the guard words become their maxima over ranks)
        # (how many updates the device has dropped so far: `skipped_step_count()` - a run whose
        # steps were silently skipped must not be reported as throughput)
        if self._skipped is None:
            self._skipped = torch.zeros(1, dtype=torch.int32, device=self.model.device)
        self._skipped += (guard[:1] != 0).to(torch.int32)
        self.model.apply_gradients(self.lr, self.beta1, self.beta2, self.eps,
                                   grad_scale=1.0 / self.world, skip=guard)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.model.device))
        self._step_done.append(done)
        return loss

    def skipped_step_count(self):
        """Training steps whose update the device dropped (guard word set: CTC status, non-finite
        loss, recurrence / weight-gradient time-out) since this trainer was built; synchronises.
        The step counter advances for such a step at first (TensorFlow's global step would not);
        `drain_checks()` takes it back out.  A dropped step always raises at the next check of
        `train_step(check=True)` / `drain_checks()`."""
        return 0 if self._skipped is None else int(self._skipped.item())

    def global_mean(self, value):
        """Average a device scalar over ranks (logged loss, eval metrics)."""
        if self.world <= 1:
            return value
        value = value.detach().clone()
        dist.all_reduce(value, op=dist.ReduceOp.SUM)
        return value / self.world
