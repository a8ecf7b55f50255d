"""Minimal counterpart of the padertorch Trainer step the reference drives
(pb_sed/experiments/weak_label_crnn/training.py:264-273,397-400):

    zero_grad -> model(batch) -> model.review -> loss.backward -> clip_grad_norm -> Adam.step

plus what the reference does not have: data parallelism.  One process per GPU; each rank runs the
full model on its shard of the minibatch and the flat fp32 gradient buffer is summed over ranks with
RCCL all-reduce (torch.distributed backend "nccl" = RCCL over xGMI), issued per bucket from inside
the backward pass as soon as a bucket's gradients are final, so the collective overlaps the
remaining conv dgrad/wgrad kernels.  Adam folds the 1/world_size average and the clip coefficient.
"""
import inspect
import time

import numpy as np
import torch
import torch.distributed as dist

from . import ops


class GradSync:
    """Bucketed, overlapped sum-all-reduce of a flat gradient buffer.

    ``buckets``: list of (start, end) element ranges; ``bucket_ready(i)`` may be called in any order
    as soon as range i is final; ``finish()`` makes the current stream wait for all collectives.
    Works with any initialised process group (gloo on CPU is what tests/test_dp_gloo.py uses).
    """

    def __init__(self, flat_grad, buckets, group=None):
        self.flat_grad, self.buckets, self.group = flat_grad, list(buckets), group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self._pending = []
        self._done = set()

    def bucket_ready(self, i):
        if self.world == 1 or i in self._done:
            return
        self._done.add(i)
        a, b = self.buckets[i]
        if b > a:
            self._pending.append(dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM,
                                                 group=self.group, async_op=True))

    def extra_sum(self, t):
        """Sum-all-reduce a small float32 tensor beside the gradient buckets (the scans' error words, Trainer._share_flags);
        complete after finish() like the buckets."""
        if self.world > 1:
            self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for i in range(len(self.buckets)):
            self.bucket_ready(i)                    # anything not announced yet
        for w in self._pending:
            w.wait()
        self._pending, self._done = [], set()
        return 1.0 / self.world


class _LibraryComm:
    """The C-ABI communicator behind LibraryGradSync (include/pbsed.h: pbsed_comm_* / pbsed_allreduce_*)."""

    def __init__(self, device):
        from . import _lib
        self._lib, self.device, self.handle = _lib, device, None

    def id_bytes(self):
        return self._lib.lib().pbsed_comm_id_bytes()

    def unique_id(self):
        import ctypes as C
        buf = C.create_string_buffer(self.id_bytes())
        self._lib.call('pbsed_comm_unique_id', buf)
        return bytes(buf.raw)

    def create(self, unique_id, rank, world):
        import ctypes as C
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            self._lib.call('pbsed_comm_create', C.c_char_p(unique_id), rank, world, C.byref(handle))
        self.handle = handle

    def begin(self, data_ptr, count):
        self._lib.call('pbsed_allreduce_begin', self.handle, data_ptr, count, self._lib.stream())

    def finish(self):
        self._lib.call('pbsed_allreduce_finish', self.handle, self._lib.stream())

    def destroy(self):
        if self.handle is not None:
            self._lib.call('pbsed_comm_destroy', self.handle)
            self.handle = None


class LibraryGradSync:
    """Same interface as GradSync, but the collective is the library's own (``pbsed_allreduce_begin`` /
    ``pbsed_allreduce_finish`` in include/pbsed.h: RCCL on a library-owned stream, event-fenced against the compute
    stream, no torch.distributed call on the data path).  The RCCL unique id is made on rank 0 and handed to the other
    ranks through whatever process group / store is initialised (a 128-byte control message, once).
    ``comm``: the communicator object (default: the C-ABI's; tests/test_dp_gloo.py passes a recording stand-in to
    check the bucket / ordering logic of this class without a GPU)."""

    def __init__(self, flat_grad, buckets, rank=None, world=None, unique_id=None, comm=None):
        self.flat_grad, self.buckets = flat_grad, list(buckets)
        if world is None:
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
            rank = dist.get_rank() if world > 1 else 0
        self.world, self.rank = world, rank
        self._comm = _LibraryComm(flat_grad.device) if comm is None else comm
        # Both stages are agreed on COLLECTIVELY when a process group is up: every rank issues the same control
        # collectives in the same order whatever fails locally (a raise on rank 0 before the broadcast would leave the
        # other ranks waiting in it; a communicator that comes up on some ranks only would leave the groups diverged).
        grouped = world > 1 and dist.is_available() and dist.is_initialized()
        err = None
        if unique_id is None:
            box = [None]
            if rank == 0:
                try:
                    box[0] = self._comm.unique_id()
                except Exception as ex:                             # ANY failure: the broadcast below must still be reached
                    err = ex
            if grouped:
                dist.broadcast_object_list(box, src=0)              # None = rank 0 could not make one
            unique_id = box[0]
            if unique_id is None:                                   # every rank sees None: every rank raises, nobody waits
                raise RuntimeError(f'library communicator: no unique id from rank 0 ({err})')
        try:
            if len(unique_id) != self._comm.id_bytes():
                raise RuntimeError(f'library communicator: unique id of {len(unique_id)} bytes, expected {self._comm.id_bytes()}')
            self._comm.create(unique_id, rank, world)
        except Exception as ex:                                     # a missing symbol, a bad id, RCCL: all reach the agreement
            err = ex
        if grouped and not self._agree(err is None):
            if err is None:                                         # up here, not everywhere: take it down again
                self._comm.destroy()
            raise RuntimeError(f'library communicator not created on every rank ({err or "another rank failed"})') from err
        if err is not None:
            raise err
        self._done = set()

    def _agree(self, ok):
        """MIN over the torch process group of a local success flag (control plane, once at construction)."""
        backend = dist.get_backend()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                            device=self.flat_grad.device if backend == 'nccl' else 'cpu')
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def bucket_ready(self, i):
        if i in self._done:
            return
        self._done.add(i)
        a, b = self.buckets[i]
        if b > a and self.world > 1:
            self._comm.begin(self.flat_grad.data_ptr() + 4 * a, b - a)

    def extra_sum(self, t):
        """See GradSync.extra_sum: one more pbsed_allreduce_begin on the library's stream, fenced by the same finish()."""
        if self.world > 1:
            self._comm.begin(t.data_ptr(), t.numel())

    def finish(self):
        for i in range(len(self.buckets)):
            self.bucket_ready(i)
        self._comm.finish()
        self._done = set()
        return 1.0 / self.world

    def close(self):
        if self._comm is not None:
            self._comm.destroy()
            self._comm = None


def make_grad_sync(flat_grad, buckets, allreduce=None):
    """The gradient exchange of a Trainer.  ``allreduce``: 'library' | 'torch' | None (env PBSED_ALLREDUCE, else the
    default).  Default: with a process group of more than one rank and the gradients on a GPU, the C-ABI's own
    communicator (LibraryGradSync - the path include/pbsed.h describes: RCCL on a library-owned stream, event fences, no
    torch call on the data path); 'torch' (torch.distributed.all_reduce on the initialised group, backend "nccl" = RCCL)
    otherwise - a single process, CPU tensors (gloo tests), or when the library's communicator cannot be created (a warning
    says so; LibraryGradSync agrees on success over the process group after each stage, so every rank takes the same branch)."""
    import os
    import warnings
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    choice = allreduce or os.environ.get('PBSED_ALLREDUCE') or ('library' if multi and flat_grad.is_cuda else 'torch')
    if choice == 'library':
        try:
            return LibraryGradSync(flat_grad, buckets), 'library'
        except (RuntimeError, OSError) as ex:
            if allreduce == 'library' or not multi:
                raise
            # LibraryGradSync agreed on the failure over the process group: every rank is here
            warnings.warn(f'library communicator unavailable ({ex}); falling back to torch.distributed.all_reduce')
    return GradSync(flat_grad, buckets), 'torch'


def shard_batch(batch, rank, world):
    """Rank r takes a contiguous run of clips of a global batch: B/world each; a remainder (a ragged last
    inference batch) goes one clip each to the first B % world ranks, so a rank's share may be empty."""
    n = len(batch['seq_len'])
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    sl = slice(start, start + base + (1 if rank < rem else 0))
    out = {}
    for k, v in batch.items():
        if isinstance(v, (torch.Tensor, np.ndarray)) and len(v) == n:
            out[k] = v[sl]
        elif isinstance(v, (list, tuple)) and len(v) == n:
            out[k] = list(v[sl])
        else:
            out[k] = v
    return out


def param_buckets(model):
    """Gradient buckets in backward-completion order: recurrent part + heads, CNN1d, CNN2d."""
    model.flat_parameters()
    spans = {}
    for name, p in model.named_parameters():
        key = 'cnn_2d' if name.startswith('cnn.cnn_2d') else 'cnn_1d' if name.startswith('cnn.cnn_1d') else 'rnn'
        a, b = spans.get(key, (p._pbsed_off, p._pbsed_off))
        spans[key] = (min(a, p._pbsed_off), max(b, p._pbsed_off + p.numel()))
    order = [k for k in ('rnn', 'cnn_1d', 'cnn_2d') if k in spans]
    return order, [spans[k] for k in order]



def load_init_checkpoint(model, state_dict):
    """Initialise a weak-label CRNN from another model's ``state_dict`` the way the reference's training script does
    (pb_sed/experiments/weak_label_crnn/training.py:327-342): CNN and both GRUs are loaded completely, of the two
    output nets everything but the last layer (whose width is the source task's number of classes).  The build's
    parameter / buffer names are the reference's (``cnn.cnn_2d.convs.<i>.conv.weight``, ``...norm.gamma|beta|
    running_mean|running_power``, ``rnn_fwd.rnn.weight_ih_l0`` ...), so a ``torch.load(path)['model']`` dict written by
    the reference's trainer is accepted as is.  Buffers the build does not keep are ignored; missing or mis-shaped
    tensors of the loaded parts raise.  A 'deep' reference model's skip convolutions (``residual_skip_convs.<s>-><d>.conv.*``,
    training.py:170-183) and padertorch's broadcast-shaped normalisation tensors are taken under this build's names / shapes
    (modules.canonical_state_dict), so `tuning.py:38`-style reuse of a 'deep' checkpoint works too."""
    own = model.state_dict()
    from .modules import canonical_state_dict
    state_dict = canonical_state_dict(state_dict, own)

    def sub(prefix):
        return {k: v for k, v in state_dict.items() if k.startswith(prefix)}

    picked = {}
    for prefix in ('cnn.', 'rnn_fwd.rnn.', 'rnn_bwd.rnn.'):
        part = sub(prefix)
        if not part and prefix == 'rnn_bwd.rnn.' and getattr(model, 'rnn_bwd', None) is None:
            continue
        missing = [k for k in own if k.startswith(prefix) and k not in part]
        if missing:
            raise KeyError(f'init checkpoint lacks {missing[:4]}{" ..." if len(missing) > 4 else ""}')
        picked.update({k: v for k, v in part.items() if k in own})
    for head in ('rnn_fwd.output_net.', 'rnn_bwd.output_net.'):
        part = sub(head)
        if not part:
            continue
        last = sorted(k[len(head):].split('.')[1] for k in part)[-1]        # 'convs.<idx>....': pop the output layer
        picked.update({k: v for k, v in part.items() if k[len(head):].split('.')[1] != last and k in own})
    for k, v in picked.items():
        if tuple(own[k].shape) != tuple(v.shape):
            raise ValueError(f'init checkpoint: {k} has shape {tuple(v.shape)}, model expects {tuple(own[k].shape)}')
    with torch.no_grad():
        for k, v in picked.items():
            own[k].copy_(v)                          # aliases the parameter storage; bumps the version the packed-weight caches key on
    return sorted(picked)

class Trainer:
    def __init__(self, model, lr=5e-4, gradient_clipping=1e10, betas=(.9, .999), eps=1e-8, allreduce=None, flag_check_lag=1):
        """``allreduce``: 'library' (the C-ABI's own RCCL communicator, LibraryGradSync: the default with more than one
        rank on GPUs) or 'torch' (torch.distributed all_reduce on the initialised process group: "nccl" = RCCL on ROCm;
        the default otherwise and the fallback); env PBSED_ALLREDUCE.  See make_grad_sync."""
        self.model = model
        self.lr, self.clip, self.betas, self.eps = lr, gradient_clipping, betas, eps
        self.flat_param, self.flat_grad = model.flat_parameters()
        self.m = torch.zeros_like(self.flat_param)
        self.v = torch.zeros_like(self.flat_param)
        self.sumsq = torch.zeros((), dtype=torch.float64, device=self.flat_param.device)
        self.grad_norm = torch.zeros((), dtype=torch.float32, device=self.flat_param.device)
        self.iteration = 0
        self.bucket_names, buckets = param_buckets(model)
        self.sync, self.allreduce = make_grad_sync(self.flat_grad, buckets, allreduce)
        model._grad_hook = self._on_grads_ready
        self._defer = 'defer_summary' in inspect.signature(model.review).parameters
        # Error words of the persistent scans.  The device side is immediate: a step whose scan timed out skips its own Adam
        # update (adam_step's skip_flags).  The HOST looks at step n's words when step n + flag_check_lag returns (default 1; the
        # words are sticky, so nothing is lost): waiting for them inside step n means waiting for the END of step n's device
        # work before the first launch of step n + 1 can be enqueued - the device then idles at every step boundary for as long
        # as the host needs to get going again (0.3 .. 0.5 ms of a 10.8 ms step; 1.4 ms on a box with a slow host).  finish()
        # looks at what is still pending; flag_check_lag=0 restores the check inside the step.
        self.flag_check_lag = flag_check_lag
        self._flags_on = self.flat_param.is_cuda       # (the emulated-device tests switch it on for CPU tensors)
        self._flags_hosts = [torch.zeros(ops.GRU_FLAG_WORDS, dtype=torch.int32).pin_memory() for _ in range(2)] \
            if self.flat_param.is_cuda else [torch.zeros(ops.GRU_FLAG_WORDS, dtype=torch.int32) for _ in range(2)]
        # Data-parallel runs: the words are AGREED ON over the ranks before Adam looks at them (_share_flags) - a hand-off that
        # timed out on one rank must make every rank skip the update and every rank raise; a rank that skipped alone would
        # leave the replicas with different parameters, and a rank that raised alone would leave the others waiting in the next
        # step's collective.
        self._flags_f32 = torch.zeros(ops.GRU_FLAG_WORDS, dtype=torch.float32, device=self.flat_param.device)
        self._flags_agreed = torch.zeros(ops.GRU_FLAG_WORDS, dtype=torch.int32, device=self.flat_param.device)
        self._flags_shared = False
        self._pending_checks = []
        self.last_enqueue_s = 0.            # host time of the last step up to (not including) the wait for its summary
        self.measure_sync, self.sync_events = False, None   # bench.py: event pairs around the wait for the collectives
        self.snapshot_statistics()
        # the baseline must be the state the replicas share when training (re)starts: a checkpoint loaded into the model
        # after this constructor would otherwise count as 'tracked since the last merge' on every rank (its history summed
        # world times).  Re-snapshot after every load_state_dict and once more at the first step.
        self._stepped = False
        self._watch_loads()

    def _watch_loads(self):
        if hasattr(self.model, 'register_load_state_dict_post_hook'):
            self.model.register_load_state_dict_post_hook(lambda module, incompatible: self.snapshot_statistics())

    def snapshot_statistics(self):
        """Record the cumulative statistics (modules with a ``num_tracked_values`` counter) as the state all replicas share
        - the baseline of the first sync_buffers() merge.  Called by __init__, after every ``model.load_state_dict`` (post
        hook) and at the first step(); call it yourself after writing the buffers any other way mid-training."""
        buffers = dict(self.model.named_buffers())
        self._synced_stats, self._synced_verified = {}, set()
        for name in buffers:
            if name.endswith('num_tracked_values'):
                prefix = name[:-len('num_tracked_values')]
                stats = [buffers[prefix + k] for k in ('running_mean', 'running_power') if prefix + k in buffers]
                self._synced_stats[prefix] = (buffers[name].detach().double().reshape(-1)[:1].clone(),
                                              [s_.detach().double().clone() for s_ in stats])

    def _on_grads_ready(self, name):
        if name in self.bucket_names:
            self.sync.bucket_ready(self.bucket_names.index(name))
            if name == 'rnn':                        # behind the BPTT scans: this step's error words are final in stream order
                self._share_flags()

    def _share_flags(self):
        """More than one rank: enqueue the sum over the ranks of this rank's scan error words (bit 0 and bit 1 kept apart:
        value = bit0 + 1024 bit1, exact in float32 for up to 1023 ranks) beside the gradient buckets - 256 bytes, overlapped
        with the CNN's backward pass like the 'rnn' bucket it rides behind.  step() decodes it after finish()."""
        if self._flags_shared or not self._flags_on or getattr(self.sync, 'world', 1) <= 1:
            return
        words = ops.gru_flags(self.flat_param.device)[0]
        self._flags_f32.copy_((words & 1) + 1024 * ((words >> 1) & 1))
        self.sync.extra_sum(self._flags_f32)
        self._flags_shared = True

    def sync_buffers(self, mode='mean'):
        """Data-parallel replicas keep their own batch-norm / feature running statistics (no SyncBN).  Before a checkpoint
        or a validation pass make them agree.  ``mode='mean'``: momentum statistics (batch norm) are averaged over the
        ranks; CUMULATIVE statistics (a module with a ``num_tracked_values`` counter: the feature extractor) are merged
        with their counts as weights - every rank contributes what it tracked SINCE the last merge (count delta and the
        matching sum deltas), so the call is idempotent: calling it twice in a row, or before every checkpoint, neither
        inflates the counter nor re-weights old data.  The baseline of the first merge is the state snapshot_statistics()
        recorded when the Trainer was built (call it again after loading a checkpoint into every replica); it is used only
        if all ranks hold the same snapshot, otherwise each rank's whole history is its delta.
        ``'rank0'`` broadcasts rank 0's buffers.  No-op without a process group."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        synced = self.__dict__.setdefault('_synced_stats', {})
        verified = self.__dict__.setdefault('_synced_verified', set())
        buffers = dict(self.model.named_buffers())
        counted = {n[:-len('num_tracked_values')] for n in buffers if n.endswith('num_tracked_values')}
        with torch.no_grad():
            for name, buf in buffers.items():
                prefix = name[:len(name) - len(name.rpartition('.')[2])]
                if not buf.is_floating_point():
                    continue
                if mode == 'rank0':
                    dist.broadcast(buf, src=0)
                elif prefix in counted:
                    continue                                 # merged below, count-weighted
                else:
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                    buf.div_(world)
            for prefix in sorted(counted):
                cnt = buffers[prefix + 'num_tracked_values']
                stats = [buffers[prefix + k] for k in ('running_mean', 'running_power') if prefix + k in buffers]
                if mode != 'rank0':
                    n = cnt.double().reshape(-1)[:1]
                    base = synced.get(prefix)
                    if prefix not in verified:
                        # First merge of this module.  The baseline is the state the replicas SHARE: the snapshot taken when
                        # the Trainer was built (replicas are built / loaded identically; snapshot_statistics()), or, without
                        # one, the current state.  Whether it really is shared is checked on the values themselves (MIN / MAX
                        # all-reduce of counter and statistics), never inferred from the counters alone: ranks that start at
                        # 0 and train on equally long clips reach EQUAL counters with DIFFERENT statistics.  Not shared ->
                        # independent histories, every rank's whole history is its delta (baseline empty).
                        cand = base if base is not None else (n.clone(), [s_.double().clone() for s_ in stats])
                        vec = torch.cat([cand[0].reshape(-1)] + [c.reshape(-1) for c in cand[1]])
                        lo, hi = vec.clone(), vec.clone()
                        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                        same = bool(torch.equal(lo, hi))
                        base = cand if same else (torch.zeros_like(n), [torch.zeros_like(s_, dtype=torch.float64) for s_ in stats])
                        verified.add(prefix)
                    base_n, base_stats = base
                    delta_n = n - base_n
                    sums = [s.double() * n - bs * base_n for s, bs in zip(stats, base_stats)]
                    dist.all_reduce(delta_n, op=dist.ReduceOp.SUM)
                    for x in sums:
                        dist.all_reduce(x, op=dist.ReduceOp.SUM)
                    total = base_n + delta_n
                    if total.item() > 0:
                        for s, bs, x in zip(stats, base_stats, sums):
                            s.copy_(((bs * base_n + x) / total).to(s.dtype))
                    cnt.copy_(total.to(cnt.dtype).reshape(cnt.shape) if cnt.numel() == 1 else total.to(cnt.dtype).expand_as(cnt))
                synced[prefix] = (cnt.double().reshape(-1)[:1].clone(), [s.double().clone() for s in stats])
            fe = getattr(self.model, 'feature_extractor', None)
            if fe is not None:                              # derived buffers follow the merged statistics
                fe.mean.copy_(fe.running_mean)
                fe.inv_std.copy_(1. / torch.sqrt((fe.running_power - fe.running_mean ** 2).clamp_min(0.) + fe.norm_eps))

    def step(self, batch):
        """One optimisation step.  Returns the review dict (loss is a device scalar, no host sync)."""
        t_start = time.perf_counter()
        if not self._stepped:
            self._stepped = True
            self.snapshot_statistics()               # whatever was loaded since __init__ (load_init_checkpoint, copy_) is shared state
        self.model.train()
        self.flat_grad.zero_()
        flags = ops.gru_flags(self.flat_param.device) if self._flags_on else None
        self._flags_shared = False
        outputs = self.model(dict(batch))
        review = self.model.review(batch, outputs, defer_summary=True) if self._defer else self.model.review(batch, outputs)
        review['loss'].backward()
        if flags is not None:
            self._share_flags()                      # (if no 'rnn' bucket announced it: frozen recurrent part)
        if self.measure_sync and self.sync_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            scale = self.sync.finish()               # the compute stream waits here for whatever is not overlapped
            e1.record()
            self.sync_events.append((e0, e1))
        else:
            scale = self.sync.finish()
        skip_words = None if flags is None else flags[0]
        if self._flags_shared:                       # the ranks' words, summed: any rank's bit is every rank's bit
            f = self._flags_f32
            self._flags_agreed.copy_((torch.remainder(f, 1024) > 0).to(torch.int32) + 2 * (f >= 1024).to(torch.int32))
            skip_words = self._flags_agreed
        self.iteration += 1
        ops.grad_sumsq(self.flat_grad, self.sumsq)
        ops.adam_step(self.flat_param, self.flat_grad, self.m, self.v, lr=self.lr, beta1=self.betas[0],
                      beta2=self.betas[1], eps=self.eps, step=self.iteration, grad_scale=scale,
                      max_norm=self.clip, sumsq=self.sumsq, norm_out=self.grad_norm,
                      skip_flags=skip_words)     # a timed-out scan (on ANY rank): no update on the device
        ops.invalidate_packed()          # parameters changed in place behind torch's version counters
        ops.refresh_packs()              # ... and every packed copy is rebuilt in one launch
        review['scalars']['grad_norm'] = self.grad_norm
        if flags is not None and flags[1]:
            host_words = self._flags_hosts[self.iteration & 1]
            host_words.copy_(skip_words, non_blocking=True)
            checked = torch.cuda.Event()
            checked.record()
            flags[1] = 0
            self._pending_checks.append((checked, host_words))
        if ops.scan_watch is not None:
            ops.scan_watch.check()               # completed event pairs only: a scan slowed down by CU contention warns
        self.last_enqueue_s = time.perf_counter() - t_start
        finalize = review.pop('_finalize', None)
        if finalize is not None:
            finalize()                               # host-side summary: waits for a copy issued after the forward pass
        self._check_flags(self.flag_check_lag)
        return review

    def _check_flags(self, keep):
        """Wait for and look at the scan error words of all but the last ``keep`` steps."""
        while len(self._pending_checks) > keep:
            checked, host_words = self._pending_checks.pop(0)
            checked.synchronize()
            ops.gru_flags_raise(host_words.numpy())

    def finish(self):
        """Look at the scan error words of the steps whose check is still pending (flag_check_lag > 0): call it before the
        results of the last steps are used - at the end of an epoch, before a checkpoint or a validation pass."""
        self._check_flags(0)

    # ------------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """What a resume needs, in the layout of a padertorch trainer checkpoint (``{'model', 'iteration', 'epoch',
        'optimizer'}``; pb_sed/experiments/weak_label_crnn/inference.py:407-413 reads ``ckpt['model']`` through
        ``Model.from_storage_dir``): the model's state_dict under the reference's parameter / buffer names and Adam's state
        as a ``torch.optim.Adam`` state_dict over ``model.parameters()`` in order (exp_avg / exp_avg_sq views of the flat moment
        buffers, ``step`` = the iteration), so that ``torch.optim.Adam.load_state_dict`` can take the 'optimizer' entry as it is.
        ('hooks', the per-hook state padertorch's trainer also writes, is carried as an empty dict: this trainer has no hooks with
        state.  Whether padertorch's ``Trainer.load_checkpoint`` accepts the file is NOT verified - padertorch is not installed.)"""
        self.finish()
        params = list(self.model.parameters())
        state = {i: {'step': torch.tensor(float(self.iteration)),
                     'exp_avg': self.m[p._pbsed_off:p._pbsed_off + p.numel()].view(p.shape).detach().clone().cpu(),
                     'exp_avg_sq': self.v[p._pbsed_off:p._pbsed_off + p.numel()].view(p.shape).detach().clone().cpu()}
                 for i, p in enumerate(params)}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'params': list(range(len(params)))}
        return {'model': {k: v.detach().clone().cpu() for k, v in self.model.state_dict().items()},
                'iteration': self.iteration, 'epoch': getattr(self, 'epoch', 0),
                'optimizer': {'state': state if self.iteration else {}, 'param_groups': [group]}, 'hooks': {}}

    def load_state_dict(self, ckpt):
        """Resume from ``state_dict()`` output or from a padertorch trainer checkpoint of the same model (Adam state keyed by
        parameter index; a checkpoint without optimizer state restarts the moments at zero)."""
        self.model.load_state_dict(ckpt['model'])                    # (the post hook re-snapshots the cumulative statistics)
        self.iteration = int(ckpt.get('iteration', 0))
        if 'epoch' in ckpt:
            self.epoch = ckpt['epoch']
        self.m.zero_(), self.v.zero_()
        opt = ckpt.get('optimizer') or {}
        params = list(self.model.parameters())
        for i, st in (opt.get('state') or {}).items():
            p = params[int(i)]
            sl = slice(p._pbsed_off, p._pbsed_off + p.numel())
            self.m[sl].copy_(st['exp_avg'].reshape(-1))
            self.v[sl].copy_(st['exp_avg_sq'].reshape(-1))
            self.iteration = max(self.iteration, int(float(st.get('step', self.iteration))))
        for g in opt.get('param_groups') or []:
            self.lr, self.betas, self.eps = g.get('lr', self.lr), tuple(g.get('betas', self.betas)), g.get('eps', self.eps)
        if self.flat_param.is_cuda:
            ops.invalidate_packed()
            ops.refresh_packs()
        self._stepped = False

    def save_checkpoint(self, storage_dir, name=None):
        """``<storage_dir>/checkpoints/ckpt_<iteration>.pth`` (padertorch's naming), after sync_buffers() in a data-parallel run;
        rank 0 writes.  Returns the path."""
        import os
        self.sync_buffers()
        path = os.path.join(storage_dir, 'checkpoints', name or f'ckpt_{self.iteration}.pth')
        if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save(self.state_dict(), path)
        return path

    def load_checkpoint(self, path, map_location='cpu', trusted_pickle=False):
        """Resume from a file ``save_checkpoint`` wrote.  Loaded with ``weights_only=True``: ``state_dict()`` holds tensors,
        numbers, tuples, lists and dicts only, so nothing in such a file needs the unpickler to construct objects - and a file
        from somewhere else cannot run code here.  ``trusted_pickle=True`` is the explicit opt-out for a legacy padertorch
        trainer checkpoint that pickled other objects (hook state, numpy scalars): only for files whose origin is known."""
        self.load_state_dict(torch.load(path, map_location=map_location, weights_only=not trusted_pickle))
