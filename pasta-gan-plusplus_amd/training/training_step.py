"""One iteration of the full-body training schedule (reference training/training_loop_fullbody.py:468-481, 604-650),
data-parallel over the GPUs of one node.

Phases, in order (each with its own Adam; lazy regularisation rescales lr/betas by interval/(interval+1)):
    Gmain (1)  Greg (G_reg_interval)  Dmain (1)  Dreg (D_reg_interval)  D_parsingmain (1)  D_parsingreg (D_reg_interval)
    D_parsingmain (1)  D_parsingreg (D_reg_interval)        -- D_parsing is listed twice in the reference (:470-471)
Per due phase: gradients are views of the phase's flat bucket (training.ddp.GradBucket.begin) -> only the phase's
module requires grad -> accumulate over the rounds of this rank's share of the batch; during the LAST backward of the
last round finished segments of the bucket are already being summed over RCCL on a side stream -> finish the exchange
-> nan_to_num -> Adam step.  A phase the loss declares statically empty (Greg: path-length regularisation is
commented out in the reference, loss_fullbody.py:200-221) is skipped on every rank alike; any other phase issues the same
collectives on every rank whatever gradients it produced (training/ddp.py), and one that produced no gradient anywhere
takes no optimizer step -- what the reference's Adam does with all-None gradients.  Then the G_ema update (:642-650).  ADA / ticks / snapshots / metrics are outside the hot path and not restated.
"""

import copy
import os
import warnings

import torch

from torch_utils.ops import _native as nat

from . import ddp


class Phase:
    def __init__(self, name, modules, opt, interval, bucket):
        self.name, self.modules, self.opt, self.interval, self.bucket = name, modules, opt, interval, bucket


class TrainingStep:
    def __init__(self, G_parts, D, D_parsing, loss, lr=0.0005, betas=(0.0, 0.99), eps=1e-8, G_reg_interval=4, D_reg_interval=16,
                 batch_size=32, ema_kimg=10, ema_rampup=None, G_ema_parts=None, graphs=False):
        """`G_parts`: dict name -> module for G_mapping / G_synthesis / G_const_encoding / G_style_encoding.
        `graphs` (single process only): every phase -- zero the bucket, forward, backward(s), nan_to_num, Adam step -- is captured into
        one hipGraph the second time it is due and replayed from then on: the ~9 000 kernel launches of an iteration stop being
        issued one by one from Python.  Measured (round 3, config 4, one MI355X): 312 ms per iteration replayed vs 304 ms eager -- outside the
        profiler the eager step is already GPU-bound (the 23-29 % idle seen under rocprofv3 is the profiler's own host cost), so this is an
        option for hosts slower than the GPU box, off by default."""
        self.G_parts, self.D, self.D_parsing, self.loss = G_parts, D, D_parsing, loss
        self.batch_size, self.ema_kimg, self.ema_rampup = batch_size, ema_kimg, ema_rampup
        self.G_ema_parts = G_ema_parts if G_ema_parts is not None else {k: copy.deepcopy(m).eval().requires_grad_(False) for k, m in G_parts.items()}
        self.all_modules = list(G_parts.values()) + [D, D_parsing]
        for m in self.all_modules:
            m.requires_grad_(False)
        self.graphs = bool(graphs) and torch.cuda.is_available() and not (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                                          and torch.distributed.get_world_size() > 1)
        self._graph = {}                 # phase index -> dict(graph, static rounds) | 'eager' (capture refused) ; filled lazily
        self._seen = set()               # phase indices that have run once eagerly (first-call work stays out of the graphs)
        adam_kw = dict(capturable=True) if self.graphs else {}
        self.phases = []
        by_module = {}                   # module set -> (bucket, first FlatAdam): the reference lists D_parsing twice (:470-471) -- two optimizers over one module

        def make_opt(params, bucket, lr_, betas_, first):
            # GPU: nan_to_num + Adam as ONE launch over flat parameter / gradient / moment buffers, per-parameter "has a gradient" flags read on the device
            # (training/flat_adam.py; PG_FLAT_ADAM=0 = torch.optim.Adam).  CPU tensors (the gloo tests): torch.optim.Adam.
            if bucket.flat.is_cuda and os.environ.get('PG_FLAT_ADAM', '1') != '0':
                from .flat_adam import FlatAdam
                return FlatAdam(bucket, lr_, betas_, eps, share_params_with=first)
            return torch.optim.Adam(params, lr=lr_, betas=betas_, eps=eps, **adam_kw)
        for name, modules, interval in (('G', list(G_parts.values()), G_reg_interval), ('D', [D], D_reg_interval),
                                        ('D_parsing', [D_parsing], D_reg_interval), ('D_parsing', [D_parsing], D_reg_interval)):
            params = [p for m in modules for p in m.parameters()]
            key = tuple(id(m) for m in modules)
            flat = params[0].is_cuda and os.environ.get('PG_FLAT_ADAM', '1') != '0'
            prev = by_module.get(key) if flat else None      # (flat route: the second optimizer of a module set shares its gradient bucket and flat parameter buffer)
            bucket = prev[0] if prev is not None else ddp.GradBucket(params)
            ratio = 1.0 if interval is None else interval / (interval + 1)
            opt = make_opt(params, bucket, lr * ratio, tuple(b ** ratio for b in betas), prev[1] if prev is not None else None)
            by_module.setdefault(key, (bucket, opt))
            if interval is None:
                self.phases.append(Phase(name + 'both', modules, opt, 1, bucket))
            else:
                self.phases.append(Phase(name + 'main', modules, opt, 1, bucket))
                self.phases.append(Phase(name + 'reg', modules, opt, interval, bucket))
        # Pre-scaled weight copies (round 5): `weight * weight_gain` of every equalised-LR convolution was two elementwise launches per layer
        # call (forward and backward), ~1 100 per iteration.  On the GPU route (gather-mode buckets) the layers take an alias of a copy that ONE multi-tensor pass
        # per optimizer step keeps current (networks._GainedAlias), and the gather multiplies the accumulated gradient by the gain.  PG_GAIN_FOLD=0: the multiplies.
        self._gained = {}                # module-set key -> (params, copies, gains)
        self._gained_table = {}
        if os.environ.get('PG_GAIN_FOLD', '1') != '0':
            from . import networks
            table = {}
            for key, (bucket, _opt) in by_module.items():
                if not (bucket.flat.is_cuda and bucket.gather):
                    continue
                index = {id(p): i for i, p in enumerate(bucket.params)}
                ps, gs = [], []
                mods = [m for ph in self.phases if ph.bucket is bucket for m in ph.modules]
                seen = set()
                for top in mods:
                    for m in top.modules():
                        if isinstance(m, networks._ConvBase) and isinstance(getattr(m, 'weight', None), torch.nn.Parameter) and id(m.weight) in index and id(m.weight) not in seen:
                            seen.add(id(m.weight))
                            ps.append(m.weight)
                            gs.append(float(m.weight_gain))
                if not ps:
                    continue
                copies = [torch.empty_like(p) for p in ps]
                gain_of = {id(p): g for p, g in zip(ps, gs)}
                # EVERY bucket over these parameters gathers gradients of the pre-scaled copies: with PG_FLAT_ADAM=0 the second D_parsing entry of the
                # phase table has a bucket of its own, and `_phase` arms the aliases for all phases alike
                for b in {id(ph.bucket): ph.bucket for ph in self.phases}.values():
                    if b.flat.is_cuda and b.gather and any(id(p) in gain_of for p in b.params):
                        b.grad_gains = [gain_of.get(id(p), 1.0) for p in b.params]
                self._gained[key] = (ps, copies, gs)
                for p, c in zip(ps, copies):
                    table[id(p)] = [c, -1, p]
            self._gained_table = table       # (armed only while one of this step's phases runs: `_phase`)
            for key in self._gained:
                self._refresh_gained(key)
        self.cur_nimg = 0
        self.batch_idx = 0
        self.observer = None             # tests: callable(event, phase) at 'begin' of a phase and when its 'gradients' are final (eager phases only)

    def _refresh_gained(self, key):
        """weight * gain of one module set into the persistent copies: two multi-tensor launches (copy, scale); records the versions the copies were made from."""
        ent = self._gained.get(key)
        if ent is None:
            return
        ps, copies, gs = ent
        with torch.no_grad():
            torch._foreach_copy_(copies, [p.detach() for p in ps])
            torch._foreach_mul_(copies, gs)
        for p in ps:
            self._gained_table[id(p)][1] = p._version

    def due_phases(self):
        return [ph for ph in self.phases if self.batch_idx % ph.interval == 0]

    def _phase(self, ph, rounds):
        """One due phase, eagerly: returns nothing; everything it does is GPU work enqueued on the current stream plus Python bookkeeping."""
        if self.observer is not None:
            self.observer('begin', ph)
        ph.bucket.begin()                                    # zero the flat bucket; every .grad is a view into it
        for m in ph.modules:
            m.requires_grad_(True)
        from . import networks
        networks._gained_provider[0] = self._gained_table if self._gained else None      # the layers' `weight * gain` = aliases of the pre-scaled copies, in this phase only
        try:
            for r, batch in enumerate(rounds):
                last = r == len(rounds) - 1
                self.loss.on_last_backward = ph.bucket.last_round if last else None
                self.loss.accumulate_gradients(phase=ph.name, sync=last, gain=ph.interval, **batch)
        finally:
            networks._gained_provider[0] = None
        self.loss.on_last_backward = None
        for m in ph.modules:
            m.requires_grad_(False)
        if not ph.bucket.finish():                           # nobody produced a gradient: nothing to exchange, nothing to step
            return
        fused = not isinstance(ph.opt, torch.optim.Optimizer)        # FlatAdam cleans the gradients in the pass that applies them
        if not fused or self.observer is not None:
            torch.nan_to_num(ph.bucket.flat, nan=0, posinf=1e5, neginf=-1e5, out=ph.bucket.flat)
        if self.observer is not None:
            self.observer('gradients', ph)                   # exchanged, cleaned gradients in place; the optimizer has not stepped yet
        ph.opt.step()
        self._refresh_gained(tuple(id(m) for m in ph.modules))   # the pre-scaled weight copies of this module set follow the step

    def _phase_graphed(self, idx, ph, rounds):
        """First time: eager (plugin loading, kernel attributes, MIOpen's solver choice stay out of the graph).  Second time: capture, then
        replay.  Later: copy the batch into the static buffers, replay."""
        if idx not in self._seen or self._graph.get(idx) == 'eager':
            self._seen.add(idx)
            self._phase(ph, rounds)
            nat.invalidate_packed_weights()
            return
        entry = self._graph.get(idx)
        if entry is None:
            static = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in rounds]
            graph = torch.cuda.CUDAGraph()
            nat.invalidate_packed_weights()                  # no pack made outside the capture may be reused inside it
            try:
                torch.cuda.synchronize()
                with torch.cuda.graph(graph):
                    self._phase(ph, static)
            except Exception as e:                           # noqa: BLE001 -- a phase that cannot be captured keeps running eagerly
                import traceback
                where = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in traceback.extract_tb(e.__traceback__)[-6:][::-1])
                warnings.warn(f'training step: phase {ph.name} is not capturable ({type(e).__name__}: {str(e).splitlines()[0]}; at {where}); it stays eager')
                self._graph[idx] = 'eager'
                torch.cuda.synchronize()
                nat.invalidate_packed_weights()
                self._phase(ph, rounds)
                nat.invalidate_packed_weights()
                return
            entry = self._graph[idx] = dict(graph=graph, static=static)
        for dst, src in zip(entry['static'], rounds):
            for k, v in src.items():
                if isinstance(v, torch.Tensor) and dst[k].data_ptr() != v.data_ptr():
                    dst[k].copy_(v)
        entry['graph'].replay()
        nat.invalidate_packed_weights()                      # the replay moved the weights without moving their version counters

    def graphed_phases(self):
        return [self.phases[i].name for i, e in sorted(self._graph.items()) if e != 'eager']

    def run(self, rounds):
        """`rounds`: this rank's accumulation rounds, each a dict of the tensors accumulate_gradients takes
        (real_img, gen_z, style_input, retain, pose, denorm_*_input, denorm_*_mask, gt_parsing)."""
        for idx, ph in enumerate(self.phases):
            if self.batch_idx % ph.interval != 0:
                continue
            if getattr(self.loss, 'phase_is_empty', lambda name: False)(ph.name):
                continue                                         # statically empty on every rank (Greg): no forward, no exchange, no step
            if self.graphs:
                self._phase_graphed(idx, ph, rounds)
            else:
                self._phase(ph, rounds)
        self._update_ema()
        self.cur_nimg += self.batch_size
        self.batch_idx += 1

    @torch.no_grad()
    def _update_ema(self):
        ema_nimg = self.ema_kimg * 1000
        if self.ema_rampup is not None:
            ema_nimg = min(ema_nimg, self.cur_nimg * self.ema_rampup)
        beta = 0.5 ** (self.batch_size / max(ema_nimg, 1e-8))
        # p_ema <- p.lerp(p_ema, beta), b_ema <- b (training_loop_fullbody.py:486-494) as multi-tensor launches: a handful per step
        # instead of two per parameter and one per buffer
        ps, ps_ema, bs, bs_ema = [], [], [], []
        for name, m in self.G_parts.items():
            ema = self.G_ema_parts[name]
            for p_ema, p in zip(ema.parameters(), m.parameters()):
                ps.append(p.detach()); ps_ema.append(p_ema)
            for b_ema, b in zip(ema.buffers(), m.buffers()):
                bs.append(b); bs_ema.append(b_ema)
        if ps:
            torch._foreach_copy_(ps_ema, torch._foreach_lerp(ps, ps_ema, beta))
        if bs:
            torch._foreach_copy_(bs_ema, bs)
