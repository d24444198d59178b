"""SWEMCore: the sequential weighted EM memory, HIP-backed.

Mirror of the reference's ``methods/SWEM/modules.py`` (same class / method / attribute names, same
argument meaning, same bank policy) so callers written against the reference
(``swem_evaluator.py:66-93``, ``swem_trainer.py:64-90``) work unchanged.  The arithmetic runs in
``libswem_hip.so``; there is no torch fallback.

Tensors cross this boundary in the reference's shapes: ``qk (B,Ck,h,w)``, ``qv (B,N,Cv,h,w)`` /
``(B,Cv,h,w)``, ``masks (B,N,2,h,w)``, bases ``kappa (B,N,2,Ck,L)``, ``nu (B,N,2,Cv,L)``,
``zita (B,N,2,1,L)``.  Feature maps produced by this package are channels-last in memory, so the
``(P,C)`` pixel-major views the kernels want are free; foreign NCHW inputs are repacked by the
transpose kernel.  The kernels take ONE frame's key and its N objects per call (the evaluator's case, B = 1); a batch of
B clips (the trainer's call shape, swem_trainer.py:59-90) is B such calls on slices -- clips share nothing in the EM.
"""
import math

import torch
from torch import nn

from . import ops


def l2norm(inp, dim):
    """modules.py:7-9 (host-side helper for random_init only)."""
    norm = torch.linalg.norm(inp, dim=dim, keepdim=True) + 1e-6
    return inp / norm


def as_nchw(t):
    """NHWC tensor -> the NCHW-shaped view the reference's callers index.  The view remembers the tensor it came from: when it
    comes back through `to_pixel_major` the ORIGINAL object is used again, with whatever a producing kernel cached on it
    (the bf16 planes of a conv output, the site its next consumers report to: ops.presplit)."""
    v = t.permute(0, 3, 1, 2)
    v.__dict__['_swem_nhwc'] = t
    return v


def to_pixel_major(t):
    """(B,C,h,w) -> contiguous (B,h,w,C) memory; free for channels-last input."""
    o = t.__dict__.get('_swem_nhwc')
    if o is not None and o.data_ptr() == t.data_ptr() and o._version == t._version and o.shape[0] == t.shape[0]:
        return o
    v = t.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    B, Cc, h, w = t.shape
    return ops.transpose(t.contiguous().view(B, Cc, h * w)).view(B, h, w, Cc)


class FeatureFusionLayer(nn.Module):
    """modules.py:13-26: parameters of the GLU fusion conv pair (executed by Engine.fuse_context)."""

    def __init__(self, indim, outdim):
        super().__init__()
        self.layer_f = nn.Conv2d(indim, outdim, kernel_size=3, stride=1, padding=1)
        self.layer_a = nn.Conv2d(indim, outdim, kernel_size=3, stride=1, padding=1)
        nn.init.orthogonal_(self.layer_f.weight.data)
        nn.init.zeros_(self.layer_f.bias.data)
        nn.init.orthogonal_(self.layer_a.weight.data)
        nn.init.zeros_(self.layer_a.bias.data)


class MemoryBank:
    """modules.py:29-60."""

    def __init__(self, mode='updated'):
        self.mode = mode
        assert (self.mode in ['fixed', 'updated'])
        self.bases = None
        self.n_objs = 0

    def initial_memory(self):
        self.bases = None
        self.n_objs = 0

    def add_new(self, bases):
        if self.bases is None:
            self.bases = bases
            self.n_objs = bases['kappa'].shape[1]
        else:
            N = bases['kappa'].shape[1]
            if N > self.n_objs:
                for key in bases.keys():
                    self.bases[key] = torch.cat([self.bases[key], bases[key][:, self.n_objs:]], dim=1)
            self.n_objs = N

    def update(self, bases):
        if self.mode == 'fixed':
            self.add_new(bases)
        else:
            self.bases = bases


class SWEMCore(nn.Module):
    """modules.py:63-310."""

    def __init__(self, n_bases=256, valdim=512, n_iters=4, tau=0.05, topl=64):
        super().__init__()
        self.n_bases = n_bases
        self.n_iters = n_iters
        self.tau = tau
        assert (self.tau > 0)
        self.valdim = valdim
        self.memories = dict()
        self.memories['first'] = MemoryBank(mode='fixed')
        self.memories['update'] = MemoryBank(mode='updated')
        self.p_drop = 0.0
        self.topl = int(min(self.n_bases, topl))
        self.fusion_layer = FeatureFusionLayer(valdim * 2 + self.topl * 2, valdim)
        self.init_on_host = False
        self._engine = None  # set by SWEM (the fusion conv needs the packed GLU weights)
        # matching's packed form of the two banks (l2-normalised keys, class-concatenated values: what modules.py:282-283,
        # 295-306 rebuild on every frame), kept current by memorize itself; `_stamp[b]` names the bank tensor (and its
        # version) bank b of the pack was built from -- any other tensor in the bank (injected by a test, restored by a
        # checkpoint, grown by a new object) makes matching repack that bank first
        self._pack = None
        self._stamp = [None, None]

    def empty(self):
        for key in self.memories.keys():
            self.memories[key].initial_memory()
        self._stamp = [None, None]       # (the pack's buffers are kept: a captured frame graph holds their addresses)

    # ------------------------------------------------------------------ packed banks
    def _pack_for(self, N, Ck, device):
        shape = (2 * N, Ck // 4 + 1, 2 * self.n_bases, 4)      # ops.new_pack: packed keys carry one extra group (norms)
        if self._pack is None or tuple(self._pack[0].shape) != shape or self._pack[0].device != device:
            self._pack = ops.new_pack(N, Ck, self.valdim, self.n_bases, device)
            self._stamp = [None, None]
        return self._pack

    @staticmethod
    def _stamp_of(bases):
        # (the last field: whether the bank's fp16 value planes were written with it -- ops.value_planes_wanted at that time)
        return (bases['kappa'], bases['kappa']._version, bases['nu'], bases['nu']._version, ops.value_planes_wanted())

    def _stamped(self, bank, bases):
        st = self._stamp[bank]
        return (st is not None and st[0] is bases['kappa'] and st[1] == bases['kappa']._version
                and st[2] is bases['nu'] and st[3] == bases['nu']._version
                and (st[4] or not ops.value_planes_wanted()))     # planes left out then, wanted now: repack

    def restamp(self):
        """Declare the pack consistent with the banks as they stand (after a frame graph moved new bases into its static
        bank tensors: the pack was updated by the same graph)."""
        for b, key in enumerate(('first', 'update')):
            bases = self.memories[key].bases
            self._stamp[b] = None if bases is None else self._stamp_of(bases)

    def repack(self):
        """Rebuild whatever part of the pack does not belong to the current bank tensors."""
        first, update = self.memories['first'].bases, self.memories['update'].bases
        assert first['kappa'].shape[0] == 1, 'the persistent pack serves one sequence (B = 1)'
        N, _, Ck, L = first['kappa'].shape[1:]
        pack = self._pack_for(N, Ck, first['kappa'].device)
        for b, bases in enumerate((first, update)):
            if bases is not None and not self._stamped(b, bases):
                ops.pack_bank(bases['kappa'].reshape(N, 2, Ck, L).contiguous(), bases['nu'].reshape(N, 2, -1, L).contiguous(),
                              pack, b)
                self._stamp[b] = self._stamp_of(bases)
        return pack

    # ------------------------------------------------------------------ single EM steps (modules.py:93-127)
    def sww_step(self, kappa, x_t, masks):
        """kappa (B,N,2,Ck,L), x_t (B,1,1,P,Ck), masks (B,N,2,P,1) -> weights (B,N,2,P,1)."""
        B, N = kappa.shape[:2]
        assert B == 1
        kn = ops.em_pack_bases(kappa.reshape(N * 2, *kappa.shape[-2:]).contiguous())
        x = x_t.reshape(x_t.shape[-2], x_t.shape[-1]).contiguous()
        w, _ = ops.em_ew(x, kn, masks.reshape(N * 2, -1).contiguous(), None, self.tau, True, False)
        return w.view(B, N, 2, -1, 1)

    def swe_step(self, x_t, kappa, weights):
        """-> z (B,N,2,P,L)."""
        B, N = kappa.shape[:2]
        assert B == 1
        L = kappa.shape[-1]
        kn = ops.em_pack_bases(kappa.reshape(N * 2, *kappa.shape[-2:]).contiguous())
        x = x_t.reshape(x_t.shape[-2], x_t.shape[-1]).contiguous()
        P = x.shape[0]
        _, z = ops.em_ew(x, kn, None, weights.reshape(N * 2, -1).contiguous(), self.tau, False, True)
        return z[:, :P].view(N, P, 2, L).permute(0, 2, 1, 3).reshape(B, N, 2, P, L)     # kernel layout: z[n][p][cls*L + l]

    def swm_step(self, z, x, kappa_, zita_):
        """z (B,N,2,P,L), x (B,1,1,Ck,P) -> kappa (B,N,2,Ck,L), zita (B,N,2,1,L)."""
        B, N, _, P, L = z.shape
        assert B == 1
        zp = torch.zeros((N, ops.em_pad(P), 2 * L), dtype=torch.float32, device=z.device)
        zp[:, :P] = z.reshape(N, 2, P, L).permute(0, 2, 1, 3).reshape(N, P, 2 * L)
        xp = x.reshape(x.shape[-2], P).t().contiguous()                          # (P, Ck) pixel-major
        kappa, zita, _ = ops.em_mstep(xp, False, zp, kappa_.reshape(N * 2, -1, L).contiguous(),
                                      zita_.reshape(N * 2, L).contiguous(), P)
        return kappa.view(B, N, 2, -1, L), zita.view(B, N, 2, 1, L)

    # ------------------------------------------------------------------ modules.py:129-168
    def swem(self, x, v, masks, bases_=None, pack=None, prior_packed=False, bank=1, out=None):
        B, Ck, H, W = x.shape
        N = masks.shape[1]
        if bases_ is None:
            kappa_, nu_, zita_ = self.random_init(size=(B, N, 2, Ck, self.n_bases), dtype=x.dtype, device=x.device)
        else:
            kappa_, nu_, zita_ = bases_['kappa'], bases_['nu'], bases_['zita']
        N_new = N - kappa_.shape[1]
        if N_new > 0:
            new_kappa, new_nu, new_zita = self.random_init(size=(B, N_new, 2, Ck, self.n_bases), dtype=x.dtype,
                                                           device=x.device)
            kappa_ = torch.cat([kappa_, new_kappa], dim=1)
            nu_ = torch.cat([nu_, new_nu], dim=1)
            zita_ = torch.cat([zita_, new_zita], dim=1)
        L = self.n_bases
        xp = to_pixel_major(x).view(B, H * W, Ck)                    # (B, P, C)
        if v.dim() == 5:                                             # (B,N,V,h,w) reference layout
            vp = to_pixel_major(v.flatten(0, 1)).view(B, N, H * W, -1)
        else:                                                        # engine output (B*N,h,w,V) NHWC
            vp = v.view(B, N, H * W, -1)
        mk = masks.reshape(B, N, 2, H * W).contiguous()
        outs = []
        for b in range(B):                                           # clips are independent (one key map per clip)
            outs.append(ops.memorize(xp[b], vp[b], mk[b], kappa_[b].reshape(N, 2, Ck, L).contiguous(),
                                     nu_[b].reshape(N, 2, -1, L).contiguous(), zita_[b].reshape(N, 2, L).contiguous(),
                                     self.n_iters, self.tau, pack=pack if B == 1 else None,
                                     prior_packed=prior_packed and N_new <= 0 and B == 1, bank=bank,
                                     out=None if (out is None or B != 1) else (
                                         out['kappa'].view(N, 2, Ck, L), out['nu'].view(N, 2, -1, L), out['zita'].view(N, 2, L))))
        if B == 1:
            kappa, nu, zita = outs[0]
            return {'kappa': kappa.view(B, N, 2, Ck, L), 'nu': nu.view(B, N, 2, -1, L), 'zita': zita.view(B, N, 2, 1, L)}
        return {'kappa': torch.stack([o[0] for o in outs]), 'nu': torch.stack([o[1] for o in outs]),
                'zita': torch.stack([o[2] for o in outs]).unsqueeze(-2)}

    def random_init(self, size, norm_dim=-2, dtype=None, device=None):
        """modules.py:170-178.  Host-side on purpose: the bases are drawn from the torch
        generator of the tensor's device exactly like the reference does, so equal seeds give equal bases."""
        B, N, _, _, L = size
        if self.init_on_host:      # parity aid: same numbers as a CPU run of the reference with the same seed
            kappa = torch.zeros(size=size, dtype=dtype)
            kappa.normal_(0, math.sqrt(2. / size[-1]))
            kappa = l2norm(kappa, dim=norm_dim).to(device)
        else:
            kappa = torch.zeros(size=size, dtype=dtype, device=device)
            kappa.normal_(0, math.sqrt(2. / size[-1]))
            kappa = l2norm(kappa, dim=norm_dim)
        nu = torch.zeros(B, N, 2, self.valdim, L, dtype=dtype, device=device)
        zita = torch.zeros(B, N, 2, 1, L, dtype=dtype, device=device) + 1e-6
        return kappa, nu, zita

    # ------------------------------------------------------------------ modules.py:183-193
    def memorize(self, qk, qv, masks):
        first, update = self.memories['first'], self.memories['update']
        frame0 = first.bases is None
        prior = first.bases if update.bases is None else update.bases
        N = masks.shape[1]
        grown = not frame0 and N > first.bases['kappa'].shape[1]       # new object ids (YouTube-VOS): the pack is rebuilt
        pack = None if (grown or qk.shape[0] != 1) else self._pack_for(N, qk.shape[1], qk.device)
        bank = 0 if frame0 else 1
        prior_packed = pack is not None and update.bases is not None and self._stamped(1, update.bases)
        # (a frame graph may name the tensors the new bases go to -- its two static state sets alternate, so no copy moves
        # the update bank back into a fixed buffer: evaluator.LookaheadGraph; consumed by this one call)
        out, self._next_out = getattr(self, '_next_out', None), None
        if out is not None and (grown or frame0 or out['kappa'].shape != prior['kappa'].shape):
            out = None
        bases = self.swem(qk, qv, masks, prior, pack=pack, prior_packed=prior_packed, bank=bank, out=out)
        if frame0:
            first.update(bases)
        else:
            first.update(bases)
            update.update(bases)
        if pack is not None:
            self._stamp[bank] = self._stamp_of(bases)
        else:
            self._stamp = [None, None]

    # ------------------------------------------------------------------ memorize in two calls (round 3)
    def memorize_begin(self, qk, masks):
        """Everything of `memorize` that does not read the value map -- the T (E, W, key M) steps, modules.py:129-163 -- so
        that a caller can run it beside the value encoder (evaluator.frame_chain).  Steady state only (both banks exist, one
        clip, no new object ids): returns a token for `memorize_end`, or None if the plain `memorize` has to be used."""
        first, update = self.memories['first'], self.memories['update']
        if first.bases is None or qk.shape[0] != 1 or masks.shape[1] != first.bases['kappa'].shape[1]:
            return None
        prior = first.bases if update.bases is None else update.bases
        B, Ck, H, W = qk.shape
        N, L = masks.shape[1], self.n_bases
        pack = self._pack_for(N, Ck, qk.device)
        prior_packed = update.bases is not None and self._stamped(1, update.bases)
        out, self._next_out = getattr(self, '_next_out', None), None
        if out is not None and out['kappa'].shape != prior['kappa'].shape:
            out = None
        xp = to_pixel_major(qk).view(H * W, Ck)
        mk = masks.reshape(N, 2, H * W).contiguous()
        kappa, zita, z = ops.memorize_keys(
            xp, mk, prior['kappa'].view(N, 2, Ck, L), prior['zita'].view(N, 2, L), self.n_iters, self.tau, pack,
            prior_packed=prior_packed, bank=1,
            out=None if out is None else (out['kappa'].view(N, 2, Ck, L), out['zita'].view(N, 2, L)))
        return {'kappa': kappa, 'zita': zita, 'z': z, 'prior': prior, 'pack': pack, 'out': out, 'shape': (N, Ck, H, W)}

    def memorize_end(self, token, qv):
        """The value update (modules.py:164-165) from the token's responsibilities, then the bank bookkeeping of `memorize`."""
        N, Ck, H, W = token['shape']
        L, prior, out = self.n_bases, token['prior'], token['out']
        if qv.dim() == 5:
            vp = to_pixel_major(qv.flatten(0, 1)).view(N, H * W, -1)
        else:
            vp = qv.view(N, H * W, -1)
        nu = ops.memorize_values(vp, token['z'], prior['nu'].view(N, 2, -1, L), prior['zita'].view(N, 2, L), token['pack'], bank=1,
                                 out=None if out is None else out['nu'].view(N, 2, -1, L))
        bases = {'kappa': token['kappa'].view(1, N, 2, Ck, L), 'nu': nu.view(1, N, 2, -1, L),
                 'zita': token['zita'].view(1, N, 2, 1, L)}
        self.memories['first'].update(bases)
        self.memories['update'].update(bases)
        self._stamp[1] = self._stamp_of(bases)

    # ------------------------------------------------------------------ modules.py:232-293
    def _affinity_readout(self, qk, first, update):
        """get_affinity + perm_inv_feat in one kernel.  qk (B,Ck,h,w) RAW (normalised in-kernel, modules.py:282-283) ->
        S (B*N,h,w,2l) and mem_out (B*N,h,w,V), NHWC, clip-major like the reference's flatten(0, 1)."""
        B, Ck, H, W = qk.shape
        xp = to_pixel_major(qk).view(B, H * W, Ck)
        N, L = first['kappa'].shape[1], first['kappa'].shape[-1]
        if B == 1 and update is not None:       # both banks: matching reads the persistent pack (kept current by memorize)
            # mem_out keeps a row pitch per object (ops.match): NHWC images, free batch stride
            mem_out, S = ops.match_packed(xp[0], self.repack(), L, self.topl, self.tau, hw=(H, W))
            return S, mem_out
        mems, Ss = [], []
        for b in range(B):     # the first matched frame of a sequence (one bank) or a batch of clips: packed in the call
            kf = first['kappa'][b].reshape(N, 2, Ck, L).contiguous()
            nf = first['nu'][b].reshape(N, 2, -1, L).contiguous()
            ku = nu = None
            if update is not None:
                ku = update['kappa'][b].reshape(N, 2, Ck, L).contiguous()
                nu = update['nu'][b].reshape(N, 2, -1, L).contiguous()
            mem_out, S = ops.match(xp[b], kf, nf, ku, nu, self.topl, self.tau)
            mems.append(mem_out)
            Ss.append(S)
        if B == 1:
            return Ss[0].view(N, H, W, -1), mems[0].unflatten(1, (H, W))
        return torch.cat(Ss).view(B * N, H, W, -1), torch.cat([m.contiguous() for m in mems]).view(B * N, H, W, -1)

    def matching(self, qk, qv):
        first, update = self.memories['first'].bases, self.memories['update'].bases
        if first is None:
            raise RuntimeError('matching before the memory was initialised')
        if update is not None and update['kappa'].shape[1] != first['kappa'].shape[1]:
            raise RuntimeError('memory banks disagree on the number of objects')
        S, mem_out = self._affinity_readout(qk, first, update)
        qvp = to_pixel_major(qv)                                      # (B,h,w,V), clip b's map shared by its objects
        if self._engine is None:
            raise RuntimeError('SWEMCore.matching needs the owning SWEM model (packed fusion weights)')
        ctx = self._engine().fuse_context(mem_out, qvp, S)           # (B*N,h,w,V) NHWC
        return as_nchw(ctx), first['kappa'].shape[1]

    def get_mem(self):
        """modules.py:295-306 (inspection only: matching reads the banks directly)."""
        kappas, nus = [], []
        for key, mem in self.memories.items():
            if mem.bases is not None:
                kappas.append(mem.bases['kappa'])
                nus.append(mem.bases['nu'])
        return torch.cat(kappas, dim=-1), torch.cat(nus, dim=-1)

    def forward(self, qk, qv):
        pass
