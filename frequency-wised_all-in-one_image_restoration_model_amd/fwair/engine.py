"""Training engine for the MI355X-native AirNet: flat parameter / gradient storage, fused Adam + MoCo EMA,
whole-step HIP-graph capture and batch-sharded data parallelism over RCCL (torch.distributed backend "nccl").

The reference trains single-GPU, eager, fp32 (train.py:73-96); this is the throughput driver SURVEY.md section 8(e,f)
asks for.  Per step (train.py:80-96 phase 2):
    zero grads -> AirNet forward (query encoder, EMA, key encoder, MoCo logits, decoder) -> L1 + w * mean CE ->
    backward -> [gradient all-reduce] -> Adam -> shadow refresh
One process per GPU; every replica keeps its own BatchNorm statistics and MoCo queue (the reference has neither
SyncBN nor a gathered queue -- net/utils/moco.py:55 is commented out), so a replica is exactly a reference run on
its shard and the only exchange is the mean of the gradients.
"""
import os

import torch
import torch.distributed as dist

from . import functional as Fn
from . import ops
from .lib import call


# ---------------------------------------------------------------------------------------------------------------
# loss  (train.py:88-92):  l1(restored, clean) + w * mean_i CE(logits_i, 0)
# ---------------------------------------------------------------------------------------------------------------
class TrainLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, restored, clean, logits, w, gscale):
        """logits: f32 [L, B, 1+K].  Returns f32 [3] = (unused, l1, contrast).  Gradients are pre-multiplied by
        gscale (1 / world size: a SUM all-reduce then yields the data-parallel mean)."""
        dev = restored.device
        restored, clean, logits = restored.contiguous(), clean.contiguous(), logits.contiguous()
        out = torch.zeros(3, dtype=torch.float32, device=dev)
        dres = torch.empty_like(restored)
        dlog = torch.empty_like(logits)
        L, B, N = logits.shape
        call('fw_l1_loss', restored, clean, dres, restored.numel(), float(gscale), out[1:2])
        call('fw_ce0_loss', logits, dlog, L * B, N, float(w) * float(gscale), out[2:3])
        ctx.save_for_backward(dres, dlog)
        ctx.w = float(w)
        return out

    @staticmethod
    def backward(ctx, dout):
        dres, dlog = ctx.saved_tensors
        # d(total) = d(l1) + w d(contrast); the kernels already folded w into dlog.  dout is (1, 0, 0) for total.
        return dres, None, dlog, None, None


class L1LossFn(torch.autograd.Function):
    """mean |restored - clean| alone (train.py:89), gradient pre-scaled by gscale."""

    @staticmethod
    def forward(ctx, restored, clean, gscale):
        restored, clean = restored.contiguous(), clean.contiguous()
        out = torch.zeros(1, dtype=torch.float32, device=restored.device)
        dres = torch.empty_like(restored)
        call('fw_l1_loss', restored, clean, dres, restored.numel(), float(gscale), out)
        ctx.save_for_backward(dres)
        return out

    @staticmethod
    def backward(ctx, dout):
        return ctx.saved_tensors[0], None, None


class ContrastLossFn(torch.autograd.Function):
    """mean_i CE(logits_i, 0) alone (phase 1 of train.py:82-86: the encoder is trained on the contrastive loss only)."""

    @staticmethod
    def forward(ctx, logits, gscale):
        logits = logits.contiguous()
        out = torch.zeros(1, dtype=torch.float32, device=logits.device)
        dlog = torch.empty_like(logits)
        L, B, N = logits.shape
        call('fw_ce0_loss', logits, dlog, L * B, N, float(gscale), out)
        ctx.save_for_backward(dlog)
        return out

    @staticmethod
    def backward(ctx, dout):
        return ctx.saved_tensors[0], None


def train_loss(restored, clean, logits, w, gscale=1.0):
    """-> (total [scalar tensor], l1, contrast); total is differentiable (its backward seeds the kernels' gradients)."""
    v = TrainLossFn.apply(restored, clean, logits, w, gscale)
    total = v[1] + w * v[2]
    return total, v[1].detach(), v[2].detach()


# ---------------------------------------------------------------------------------------------------------------
# flat storage
# ---------------------------------------------------------------------------------------------------------------
def _layout(params):
    """Element offset of every parameter in a flat buffer; each starts on a multiple of 8 elements, i.e. 16-byte aligned
    both in the f32 buffer and in its bf16 shadow (vector loads of biases, GEMM operands) -- except a tensor marked
    `_fw_pack` (ordered_parameters), which continues its predecessor with no padding so the group is ONE dense array."""
    offs, o, end = [], 0, 0
    for p in params:
        start = end if getattr(p, '_fw_pack', False) else o
        offs.append(start)
        end = start + p.numel()
        o = (end + 7) // 8 * 8
    return offs, o


def ordered_parameters(root):
    """root.parameters() with each LinearProjection's tensors regrouped as (to_q.weight, to_kv.weight, to_q.bias, to_kv.bias):
    adjacent in a flat buffer they form one [3C, C] weight and one [3C] bias."""
    from .modules import FrequencyWindowAttention, LinearProjection
    group = {}
    for mod in root.modules():
        if isinstance(mod, LinearProjection) and mod.to_q.bias is not None:
            g = [mod.to_q.weight, mod.to_kv.weight, mod.to_q.bias, mod.to_kv.bias]
            for t in g:
                group[id(t)] = g
        elif isinstance(mod, FrequencyWindowAttention):
            # the L*L relative-position tables of one attention (encoder_Uformer.py:216-219) packed back to back: the kernel's
            # [L*L, 225, heads] operand and its gradient become plain views (no stack forward, no 9 accumulate-adds backward)
            for t in list(mod.relative_position_bias_table)[1:]:
                t._fw_pack = True
    out, seen = [], set()
    for p in root.parameters():
        for t in group.get(id(p), [p]):
            if id(t) not in seen:
                seen.add(id(t))
                out.append(t)
    return out


def flatten_parameters(params, device=None):
    """Re-home `params` into one contiguous, zero-padded f32 buffer (each parameter becomes a 16-byte aligned view)."""
    params = list(params)
    device = device or params[0].device
    offs, n = _layout(params)
    flat = torch.zeros(n, dtype=torch.float32, device=device)
    with torch.no_grad():
        for p, o in zip(params, offs):
            v = flat[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
    return flat


def attach_flat_grads(params, flat_g):
    offs, _ = _layout(params)
    for p, o in zip(params, offs):
        p.grad = flat_g[o:o + p.numel()].view_as(p)


class GradAllReducer:
    """Mean of the flat gradient buffer over the data-parallel group, in a few large buckets (xGMI is
    point-to-point: few, large collectives).  Works with any backend (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, flat_g, group=None, bucket_elems=64 * 1024 * 1024, wire_dtype=torch.float32):
        self.flat_g, self.group, self.wire_dtype = flat_g, group, wire_dtype
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = flat_g.numel()
        self.buckets = [(s, min(n, s + bucket_elems)) for s in range(0, n, bucket_elems)]
        self._wire = None
        if wire_dtype != torch.float32:
            self._wire = torch.empty(min(n, bucket_elems), dtype=wire_dtype, device=flat_g.device)

    def launch(self, lo, hi):
        """Asynchronous SUM all-reduce of flat_g[lo:hi] in buckets; returns the work handles (wait with `finish`)."""
        works = []
        if self.world == 1 or hi <= lo:
            return works
        step = self.buckets[0][1] - self.buckets[0][0]
        for s in range(lo, hi, step):
            g = self.flat_g[s:min(hi, s + step)]
            if self._wire is None:
                works.append((dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, g))
            else:                                             # low-precision wire: one staging buffer per bucket in flight
                w = torch.empty(g.numel(), dtype=self.wire_dtype, device=g.device)
                w.copy_(g)
                works.append((dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group, async_op=True), w, g))
        return works

    @staticmethod
    def finish(works):
        for wk, w, g in works:
            wk.wait()
            if w is not None:
                g.copy_(w)

    def __call__(self, upto=None):
        """upto: only the first `upto` elements carry gradients (phase 1: the query-encoder slice)."""
        if self.world == 1:
            return
        works = []
        for s, e in self.buckets:
            if upto is not None:
                if s >= upto:
                    break
                e = min(e, upto)
            g = self.flat_g[s:e]
            if self._wire is None:
                works.append((dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, g))
            else:
                w = self._wire[:e - s]
                w.copy_(g)
                dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
                g.copy_(w)
        for wk, _, _ in works:
            wk.wait()                                  # gradients were pre-scaled by 1/world in the loss


# ---------------------------------------------------------------------------------------------------------------
# engine
# ---------------------------------------------------------------------------------------------------------------
class TrainEngine:
    def __init__(self, net, lr=2e-4, contrast_loss_weight=0.6, betas=(0.9, 0.999), eps=1e-8, use_graph=True,
                 grad_wire_dtype=torch.float32, split_backward=False, freq_l1=None):
        self.net = net
        self.freq_l1 = freq_l1                             # (FrequencyDecompose(inverse=False), weight): the optional frequency L1 term of train.py:69-70,90-91
        self.split_backward = split_backward               # force the two-stage backward on a single GPU too (tests)
        self._split = self._gsplit = None
        self.w = float(contrast_loss_weight)
        self.betas, self.eps = betas, eps
        self.use_graph = use_graph
        moco = net.E.E
        enc_q = ordered_parameters(moco.encoder_q)
        enc_k = ordered_parameters(moco.encoder_k)
        self._k_ids = {id(p) for p in enc_k}
        qids = {id(q) for q in enc_q}
        rest = [p for p in ordered_parameters(net) if p.requires_grad and id(p) not in qids]
        self.trainable = enc_q + rest                      # query encoder first: its slice mirrors the key encoder
        dev = enc_q[0].device
        assert dev.type == 'cuda', 'TrainEngine needs the HIP device'
        self.flat_p = flatten_parameters(self.trainable, dev)
        self.flat_k = flatten_parameters(enc_k, dev)
        if dist.is_initialized() and dist.get_world_size() > 1:
            # identical replicas at start -- BEFORE the bf16 operand shadows are cast below, so that no rank's first
            # forward runs on the shadow of its own random initialisation
            dist.broadcast(self.flat_p, src=0)
            dist.broadcast(self.flat_k, src=0)
        self.n_enc = _layout(enc_q)[1]                       # padded extents: the two encoder buffers share one layout
        self.n = self.flat_p.numel()
        assert self.flat_k.numel() == self.n_enc
        self.flat_g = torch.zeros_like(self.flat_p)
        attach_flat_grads(self.trainable, self.flat_g)
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        # Adam step counts: torch.optim.Adam (train.py:63) creates a parameter's state at its FIRST gradient, so after the
        # encoder-only epochs the decoder's bias correction starts from step 1 -- two hyper vectors {lr, beta1^t, beta2^t, t}
        self.hyper = torch.tensor([lr, 1.0, 1.0, 0.0], dtype=torch.float32, device=dev)          # query encoder segment [0, n_enc)
        self.hyper_rest = torch.tensor([lr, 1.0, 1.0, 0.0], dtype=torch.float32, device=dev)     # everything else [n_enc, n)
        # bf16 operand shadows of every 2-D weight whose rows stay 16-byte aligned: written by the Adam / EMA kernels
        self.shadow_p = self.shadow_k = None
        if Fn.config.compute_dtype == torch.bfloat16:
            self.shadow_p = torch.empty(self.n, dtype=torch.bfloat16, device=dev)
            self.shadow_k = torch.empty(self.n_enc, dtype=torch.bfloat16, device=dev)
            call('fw_cast_flat', 1, self.flat_p, self.shadow_p, self.n)
            call('fw_cast_flat', 1, self.flat_k, self.shadow_k, self.n_enc)
            for plist, sh in ((self.trainable, self.shadow_p), (enc_k, self.shadow_k)):
                offs, _ = _layout(plist)
                for p, o in zip(plist, offs):
                    if p.dim() == 2 and (p.shape[1] * 2) % 16 == 0:
                        p._fw_shadow = sh[o:o + p.numel()].view_as(p)
        self._fuse_projections(net)
        self._fuse_tables(net)
        moco._ema_hook = self._ema
        self.allreduce = GradAllReducer(self.flat_g, wire_dtype=grad_wire_dtype)
        Fn.config.shadow_epoch += 1
        Fn.config.direct_grads = True                     # kernels add into the flat .grad views; autograd sees None
        self._graph = None
        self._static = None
        self.last = None

    def _fuse_projections(self, net):
        """to_q / to_kv of a LinearProjection lie back to back in the flat buffers (ordered_parameters): expose them as ONE
        [3C, C] weight / [3C] bias (+ gradient and bf16 shadow views) so that QKV is one GEMM forward and two backward."""
        from .modules import LinearProjection
        for flat, grad, shadow in ((self.flat_p, self.flat_g, self.shadow_p), (self.flat_k, None, self.shadow_k)):
            base = flat.data_ptr()
            lo, hi = base, base + flat.numel() * 4
            for mod in net.modules():
                if not isinstance(mod, LinearProjection) or mod.to_q.bias is None:
                    continue
                wq, wkv, bq, bkv = mod.to_q.weight, mod.to_kv.weight, mod.to_q.bias, mod.to_kv.bias
                C, K = wq.shape
                if not (lo <= wq.data_ptr() < hi) or C % 8 or wkv.shape != (2 * C, K):
                    continue
                if wq.data_ptr() + wq.numel() * 4 != wkv.data_ptr() or bq.data_ptr() + bq.numel() * 4 != bkv.data_ptr():
                    continue
                ow, ob = (wq.data_ptr() - base) // 4, (bq.data_ptr() - base) // 4
                fused = {'w': flat[ow:ow + 3 * C * K].view(3 * C, K), 'b': flat[ob:ob + 3 * C]}
                if grad is not None:
                    fused['gw'], fused['gb'] = grad[ow:ow + 3 * C * K].view(3 * C, K), grad[ob:ob + 3 * C]
                if shadow is not None and (K * 2) % 16 == 0:
                    fused['sw'] = shadow[ow:ow + 3 * C * K].view(3 * C, K)
                mod._fw_fused = fused

    def _fuse_tables(self, net):
        """Dense [L*L, 225, heads] views over the packed relative-position tables of every FrequencyWindowAttention."""
        import types
        from .modules import FrequencyWindowAttention
        for flat, grad in ((self.flat_p, self.flat_g), (self.flat_k, None)):
            base = flat.data_ptr()
            lo, hi = base, base + flat.numel() * 4
            for mod in net.modules():
                if not isinstance(mod, FrequencyWindowAttention):
                    continue
                tabs = list(mod.relative_position_bias_table)
                if not (lo <= tabs[0].data_ptr() < hi):
                    continue
                n = tabs[0].numel()
                if any(t.data_ptr() != tabs[0].data_ptr() + i * n * 4 for i, t in enumerate(tabs)):
                    continue
                o = (tabs[0].data_ptr() - base) // 4
                shape = (len(tabs),) + tuple(tabs[0].shape)
                mod._fw_tab = flat[o:o + len(tabs) * n].view(shape)
                mod._fw_tab_grad = types.SimpleNamespace(grad=grad[o:o + len(tabs) * n].view(shape)) if grad is not None else None

    def set_lr(self, lr):
        self.hyper[0:1].fill_(float(lr))
        self.hyper_rest[0:1].fill_(float(lr))

    def optimizer_state(self):
        """Everything a resume needs beside the model's state_dict (flat moments + step counters, host tensors)."""
        return {'m': self.m.cpu(), 'v': self.v.cpu(), 'hyper': self.hyper.cpu(), 'hyper_rest': self.hyper_rest.cpu()}

    def load_optimizer_state(self, st):
        self.m.copy_(st['m']); self.v.copy_(st['v'])
        self.hyper.copy_(st['hyper']); self.hyper_rest.copy_(st['hyper_rest'])

    def resync(self):
        """After load_state_dict(): the parameters changed under the engine -- refresh the bf16 operand shadows."""
        if self.shadow_p is not None:
            call('fw_cast_flat', 1, self.flat_p, self.shadow_p, self.n)
            call('fw_cast_flat', 1, self.flat_k, self.shadow_k, self.n_enc)
        Fn.config.shadow_epoch += 1
        Fn.refresh_shadows('all')         # the captured forward reads these buffers without re-deriving them (the graph's own refresh
                                          # follows its Adam step)

    def _ema(self):
        call('fw_ema', 1 if self.shadow_k is not None else 0, self.flat_k, self.flat_p, self.shadow_k, self.n_enc, self.net.E.E.m)
        Fn.config.shadow_epoch += 1
        Fn.refresh_shadows('key', only_ids=self._k_ids, restamp_others=True)      # only the key encoder's parameters changed

    def _freq_term(self, restored, clean):
        """weight * L1(decompose(restored), decompose(clean)), pre-scaled like the other loss gradients (differentiable module)."""
        dec, wgt = self.freq_l1
        d = (dec(restored.float()) - dec(clean)).abs().mean()
        return d * wgt

    def _fwd_bwd(self, xq, xk, clean):
        self.flat_g.zero_()
        restored, logits, labels = self.net(x_query=xq, x_key=xk)
        total, l1, contrast = train_loss(restored, clean, torch.stack(logits, 0), self.w, 1.0 / self.allreduce.world)
        if self.freq_l1 is not None:
            f = self._freq_term(restored, clean)
            (total + f / self.allreduce.world).backward()
            return torch.stack([(total + f).detach(), l1 + f.detach(), contrast])
        total.backward()
        return torch.stack([total.detach(), l1, contrast])

    # ---- backward in two stages (data parallel): the decoder's gradients are complete when the decoder's backward is, long before
    # the encoder's -- their all-reduce runs on RCCL's stream while the encoder backward is still computing
    def _split_a(self, xq, xk, clean):
        """zero grads, forward, L1 loss, backward of the DECODER (incl. the lambda heads) down to the encoder's output stack."""
        self.flat_g.zero_()
        Fn.droppath_begin(xq.device, 'airnet')
        _, logits, _, inter = self.net.E(xq, xk, True)
        stack = inter[0]._fw_stack
        # cut the tape at the encoder's output: the decoder sees a detached leaf, so `backward` runs ALL of its nodes (a gradient
        # restricted to `inputs=[stack]` would prune every node that only feeds parameters) and leaves d(stack) in the leaf's .grad
        cut = stack.detach().requires_grad_(True)
        inter_d = tuple(cut.unbind(0))
        inter_d[0]._fw_stack = cut
        restored = self.net.R(xq, inter_d)
        gs = 1.0 / self.allreduce.world
        l1 = L1LossFn.apply(restored, clean, gs)
        if self.freq_l1 is not None:
            f = self._freq_term(restored, clean)
            (l1 + f * gs).backward()
            l1 = l1.detach() + f.detach()
        else:
            l1.backward()                                    # ends with the fold of the decoder's split partials (ops.flush_slabs)
        self._split = (stack, cut.grad, torch.stack(logits, 0), l1.detach())

    def _split_b(self):
        """contrastive loss, backward of the query encoder (heads + body) with the decoder's gradient of the stack added in."""
        stack, dstack, logits, l1 = self._split
        c = ContrastLossFn.apply(logits, self.w / self.allreduce.world)
        torch.autograd.backward([stack, c], [dstack, torch.ones_like(c)])
        self._split = None
        cd = c.detach()
        return torch.cat([l1 + self.w * cd, l1, cd])

    def step_split_eager(self, xq, xk, clean):
        self._split_a(xq, xk, clean)
        wa = self.allreduce.launch(self.n_enc, self.n)
        out = self._split_b()
        wb = self.allreduce.launch(0, self.n_enc)
        GradAllReducer.finish(wa); GradAllReducer.finish(wb)
        self._optim()
        self.last = out
        return out

    def _capture_split(self, warmup=2):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.step_split_eager(*self._static)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ops.reserve_capture_tables()
        ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            self._split_a(*self._static)
        with torch.cuda.graph(gb, pool=ga.pool()):
            self._out = self._split_b()
        with torch.cuda.graph(gc, pool=ga.pool()):
            self._optim()
        self._gsplit = (ga, gb, gc)

    def _optim(self, phase=2):
        sh = 1 if self.shadow_p is not None else 0
        ne = self.n_enc
        call('fw_adam_tick', self.hyper, self.betas[0], self.betas[1])
        call('fw_adam', sh, self.flat_p, self.flat_g, self.m, self.v, self.shadow_p, ne, self.hyper, self.betas[0], self.betas[1], self.eps)
        if phase == 2 and self.n > ne:
            call('fw_adam_tick', self.hyper_rest, self.betas[0], self.betas[1])
            call('fw_adam', sh, self.flat_p[ne:], self.flat_g[ne:], self.m[ne:], self.v[ne:],
                 self.shadow_p[ne:] if self.shadow_p is not None else None, self.n - ne, self.hyper_rest, self.betas[0], self.betas[1], self.eps)
        Fn.config.shadow_epoch += 1
        Fn.refresh_shadows('all')                            # re-laid-out operand copies of the new weights, one launch

    # ---- phase 1 (train.py:82-86): encoder only, contrastive loss only -------------------------------------------------
    def _fwd_bwd_p1(self, xq, xk):
        self.flat_g[:self.n_enc].zero_()
        _, logits, _, _ = self.net.E(x_query=xq, x_key=xk)
        loss = ContrastLossFn.apply(torch.stack(logits, 0), 1.0 / self.allreduce.world)
        loss.backward()
        return loss.detach()

    def step_phase1(self, xq, xk):
        """One encoder-only step.  Captured into its own HIP graphs on first use (same replay scheme as `step`)."""
        if not self.use_graph:
            out = self._fwd_bwd_p1(xq, xk)
            self.allreduce(self.n_enc)
            self._optim(1)
            return out
        if getattr(self, '_p1', None) is None:
            static = (xq.clone(), xk.clone())
            snap = self._snapshot()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self._fwd_bwd_p1(*static); self.allreduce(self.n_enc); self._optim(1)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            single = self.allreduce.world == 1
            ops.reserve_capture_tables()
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                out = self._fwd_bwd_p1(*static)
                if single:
                    self._optim(1)
            g2 = None
            if not single:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g1.pool()):
                    self._optim(1)
            self._restore(snap)
            self._p1 = (static, g1, g2, out)
        static, g1, g2, out = self._p1
        if xq is not static[0]:
            static[0].copy_(xq, non_blocking=True); static[1].copy_(xk, non_blocking=True)
        g1.replay()
        if g2 is not None:
            self.allreduce(self.n_enc)
            g2.replay()
        Fn.config.shadow_epoch += 1
        return out

    def step_eager(self, xq, xk, clean):
        out = self._fwd_bwd(xq, xk, clean)
        self.allreduce()
        self._optim()
        self.last = out
        return out

    def _snapshot(self):
        """Model + optimizer state before the warm-up steps a capture needs (they are real steps: lazily built tables, shadow
        caches and the DropPath plan must exist before a graph can be recorded) -- restored afterwards, so that the first replay
        is the first training step."""
        return ({k: v.clone() for k, v in self.net.state_dict().items()}, self.m.clone(), self.v.clone(), self.hyper.clone(),
                self.hyper_rest.clone())

    def _restore(self, snap):
        sd, m, v, h, hr = snap
        self.net.load_state_dict(sd)                        # in-place copies: the flat views stay where they are
        self.m.copy_(m); self.v.copy_(v); self.hyper.copy_(h); self.hyper_rest.copy_(hr)
        self.resync()

    def capture(self, xq, xk, clean, warmup=2):
        """Warm up eagerly on a side stream, then capture forward+backward (and, single-GPU, the optimizer) into HIP graphs."""
        self._static = (xq.clone(), xk.clone(), clean.clone())
        snap = self._snapshot()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.step_eager(*self._static)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        single = self.allreduce.world == 1
        self._gsplit = None
        if self.split_backward or not single:
            ok = 1
            try:
                self._capture_split(warmup)
            except RuntimeError as e:                        # HIP / capture errors only: never lose the run to the overlap
                import warnings
                warnings.warn(f'fwair: two-stage backward capture failed ({type(e).__name__}: {e}); all-reduce will follow the backward pass')
                ok = 0
                self._split = None
                torch.cuda.synchronize()
            if not single:
                # the fallback issues ONE whole-buffer all-reduce per step, the split scheme two partial ones: every rank must take
                # the same path or the collectives mismatch and the job hangs -- agree on the minimum of the success flags
                flag = torch.tensor([ok], dtype=torch.int32, device=self.flat_p.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok:
                self._restore(snap)
                self._graph = True
                return
            self._gsplit = None
        ops.reserve_capture_tables()
        self._g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g1):
            self._out = self._fwd_bwd(*self._static)
            if single:
                self._optim()
        self._g2 = None
        if not single:
            self._g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g2, pool=self._g1.pool()):
                self._optim()
        self._restore(snap)
        self._graph = True

    def step(self, xq, xk, clean):
        if not self.use_graph:
            return self.step_eager(xq, xk, clean)
        if self._graph is None:
            self.capture(xq, xk, clean)
        sq, sk, sc = self._static
        if xq is not sq:
            sq.copy_(xq, non_blocking=True); sk.copy_(xk, non_blocking=True); sc.copy_(clean, non_blocking=True)
        if self._gsplit is not None:
            ga, gb, gc = self._gsplit
            ga.replay()
            wa = self.allreduce.launch(self.n_enc, self.n)    # decoder gradients travel while the encoder backward runs
            gb.replay()
            wb = self.allreduce.launch(0, self.n_enc)
            GradAllReducer.finish(wa); GradAllReducer.finish(wb)
            gc.replay()
            Fn.config.shadow_epoch += 1                      # the replay changed the weights behind functional.shadow()'s stamps
            self.last = self._out
            return self._out
        self._g1.replay()
        if self._g2 is not None:
            self.allreduce()
            self._g2.replay()
        Fn.config.shadow_epoch += 1
        self.last = self._out
        return self._out


def init_distributed():
    """torchrun-style environment -> (rank, local_rank, world).  Backend "nccl" is RCCL on ROCm."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # rehearsal aids (a one-GPU box): FW_DIST_DEVICE pins every rank to one device, FW_DIST_BACKEND=gloo replaces RCCL (which refuses
    # two ranks on one device) -- the engine's multi-rank control flow (three-graph capture, overlapped bucket all-reduce, the ranks'
    # agreement on the fallback) then runs for real, only the wire is different
    if 'FW_DIST_DEVICE' in os.environ:
        local = int(os.environ['FW_DIST_DEVICE'])
    backend = os.environ.get('FW_DIST_BACKEND', 'nccl')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
