"""Shared helpers for the parity tests (oracle side + clip / weight regeneration)."""
import torch

from oracle import swem_oracle as O
from swem_amd import synth, weights
from swem_amd.swem import SWEM


def make_model_and_sd(cfg, wseed, device=None, pred_scale=None):
    """Product model with seeded weights (+ the same state dict on CPU for the oracle).  pred_scale: the training
    fixtures shrink the prediction head (tests/golden/make_golden_train.py)."""
    model = SWEM(cfg)
    sd = weights.fill_state_dict(model.state_dict(), seed=wseed, backbone=cfg.BACKBONE)
    if pred_scale is not None:
        sd['decoder.pred.weight'] = sd['decoder.pred.weight'] * pred_scale
        sd['decoder.pred.bias'] = sd['decoder.pred.bias'] * 0
    model.load_state_dict(sd, strict=True)
    model.eval()
    # the tests pick their arithmetic EXPLICITLY (helpers.arith): a test model starts on the exact fp32 kernels, not on the
    # product's default for untuned shapes (ops.MODEL_FALLBACK = f16x3 on the heuristic tile; test_product_defaults covers that)
    model.book.fallback = 0
    if device is not None:
        model = model.to(device)
        model.swem_core.init_on_host = True    # same random bases as a CPU run with the same seed
    return model, sd


def clip_from_fixture(fx):
    frames, m0 = synth.make_clip(t=int(fx['t']), h=int(fx['h']), w=int(fx['w']), n_obj=int(fx['n_obj']),
                                 out_hw=(int(fx['out_h']), int(fx['out_w'])), seed=int(fx['seed']))
    return frames, m0


ROUND = 'r06'
ARITH_MODES = ('fp32', 'f16x3', 'bf16x3', 'tuned')


def tuned_plans_path():
    """The plan file of the bench configuration that ships with the library (what `python bench.py` runs by default)."""
    import os
    from swem_amd import ops
    path = ops.shipped_plans()
    return path if os.path.exists(path) else None


class arith:
    """Context: the conv arithmetic a parity test runs, selected EXPLICITLY and checked afterwards.
      'fp32'   -- no plans, no forced math: every GEMM on the fp32 MFMA (plan 0).
      'f16x3'  -- ops.conv_math((7,)): every convolution the pre-split kernel can take runs on the fp16 (hi, mid) planes (22-23
                  significant bits per operand, three products: fp32-level error), the heuristic tile; matching's value readout
                  on its bf16 planes (the pack's format).
      'bf16x3' -- ops.conv_math((3,)): every convolution the pre-split kernel can take AND matching's value readout run on
                  the hi + mid bf16 planes (16 significant bits per operand, three products), the heuristic tile.
      'tuned'  -- the bench's plans (swem_amd/plans/mi355x_480p_k256.json, what ships) loaded into the model's book: the mix of tiles, K-splits
                  and math modes the tuner chose at config B (f16x3 nearly everywhere; other shapes find no plan and run fp32).
    `ran` = {math field: conv launches} of what really ran; leaving the context asserts it matches the mode."""

    def __init__(self, mode, *models, need_bf16x3=True):
        from swem_amd import ops
        self.ops, self.mode, self.models, self.need = ops, mode, models, need_bf16x3
        self.ran, self.cm = {}, None

    def __enter__(self):
        ops = self.ops
        assert self.mode in ARITH_MODES and ops.MATH_RAN is None
        if self.mode in ('bf16x3', 'f16x3'):
            self.cm = ops.conv_math((3,) if self.mode == 'bf16x3' else (7,))
            self.cm.__enter__()
        elif self.mode == 'tuned':
            path = tuned_plans_path()
            assert path, 'no shipped plan file (swem_amd/plans/)'
            for m in self.models:              # models, or PlanBooks
                getattr(m, 'book', m).load(path)
        ops.MATH_RAN = self.ran
        return self

    def __exit__(self, et, ev, tb):
        self.ops.MATH_RAN = None
        if self.cm is not None:
            self.cm.__exit__(et, ev, tb)
        if et is None:
            total = sum(self.ran.values())
            assert total > 0, 'no conv launch was counted'
            if self.mode == 'fp32':
                assert set(self.ran) == {0}, self.ran
            elif self.mode in ('bf16x3', 'f16x3'):
                # everything but the layers the pre-split kernel cannot take (7x7 stems on 4 / 8 channels, 1-channel heads)
                m, other = (3, 7) if self.mode == 'bf16x3' else (7, 3)
                assert self.ran.get(m, 0) >= 0.85 * total and not self.ran.get(1) and not self.ran.get(2) and not self.ran.get(other), self.ran
            elif self.need:
                assert self.ran.get(7, 0) > 0 and not self.ran.get(3), self.ran      # the shipped plans: f16x3, no 16-bit layer
        return False

    def summary(self):
        names = {0: 'fp32', 1: 'bf16x6', 2: 'bf16', 3: 'bf16x3', 7: 'f16x3'}
        return {names[k]: v for k, v in sorted(self.ran.items())}


def record_parity(key, data):
    """Measured parity numbers of the GPU run -> gpurun_out/rNN_parity.json (merged key by key; copied to
    profiles/rNN_parity.json and committed from the round's own GPU run).  Never fails a test."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', ROUND + '_parity.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        cur = {}
        if os.path.exists(path):
            with open(path) as f:
                cur = json.load(f)
        cur[key] = data
        with open(path, 'w') as f:
            json.dump(cur, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


def checksum(t):
    return float(t.double().abs().sum())


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def structured_keys(P, C, n_clusters, g, noise=0.15):
    """Clustered keys whose cluster is independent of the pixel position (see tests/golden/make_golden.py)."""
    centres = torch.randn(n_clusters, C, generator=g)
    assign = torch.randint(0, n_clusters, (P,), generator=g)
    return centres[assign] + noise * torch.randn(P, C, generator=g), assign


def em_inputs(h, w, C, V, N, g):
    """Clustered keys, random values, soft rectangular fg/bg masks for N objects."""
    P = h * w
    xk, _ = structured_keys(P, C, 6, g)
    x = xk.t().reshape(1, C, h, w).contiguous()
    v = torch.randn(1, N, V, h, w, generator=g)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    fg = torch.stack([((xx >= (n * w) // (N + 1)) & (xx < ((n + 1) * w) // (N + 1)) & (yy >= h // 5)).float().flatten()
                      for n in range(N)])
    sf = (fg * 0.9 + 0.1 * torch.rand(N, P, generator=g)).clamp(0, 1)
    m = torch.stack([(1 - fg) * (1 - sf), fg * sf], 1).view(1, N, 2, h, w)
    return x, v, m


def ytvos_masks(per_frame, appear_at):
    """Object 1 annotated at frame 0, object 2 only from frame `appear_at` on (same construction as make_golden.py)."""
    m0 = per_frame[0]
    first = torch.stack([1 - m0[:, 1], m0[:, 1]], 1)
    late = torch.stack([torch.zeros_like(per_frame[appear_at][:, 2]), per_frame[appear_at][:, 2]], 1)
    masks = [first] + [None] * (len(per_frame) - 1)
    masks[appear_at] = late
    return masks


class SeededInit:
    """Re-seeds the torch generator before every 'init' (each TTA pass of the fixture run was seeded the same way)."""

    def __init__(self, model, seed, gpu_core=None):
        self.model, self.seed = model, seed

    def __call__(self, mode, *a):
        if mode == 'init':
            torch.manual_seed(self.seed)
        return self.model(mode, *a)

    def __getattr__(self, name):          # (book, swem_core, ...: what the evaluator loops look at besides calling the model)
        return getattr(self.__dict__['model'], name)


def train_cases():
    """Training-step cases and configs recorded by tests/golden/make_golden_train.py."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g9_train_cases.json')) as f:
        return json.load(f)


def train_batch(case):
    """Same construction as make_golden_train.make_batch: frames (B,T,3,H,W), init one-hot mask (B,N+1,H,W),
    labels (B,T,H,W) int64, valid_obj (B,N+1)."""
    h, w = case['hw']
    fr, im, lb = [], [], []
    for i in range(case['b']):
        frames, per = synth.make_clip(t=case['t'], h=h, w=w, n_obj=case['n'], out_hw=(h, w), seed=case['seed'] + i,
                                      all_masks=True)
        valid = torch.tensor(case['valid'][i], dtype=torch.float32)
        lab = torch.stack([m[0].argmax(0) for m in per])
        for o in range(1, case['n'] + 1):
            if valid[o] < 0.5:
                lab[lab == o] = 0
        m0 = torch.stack([(lab[0] == o).float() for o in range(case['n'] + 1)])
        fr.append(frames[0])
        im.append(m0)
        lb.append(lab)
    return torch.stack(fr), torch.stack(im), torch.stack(lb), torch.tensor(case['valid'], dtype=torch.float32)


def trainable_sd(sd, model):
    """Oracle-side state dict whose parameters (not the frozen-BN running statistics) require grad."""
    names = {k for k, _ in model.named_parameters()}
    out = {}
    for k, v in sd.items():
        t = v.clone()
        if k in names and t.dtype.is_floating_point:
            t.requires_grad_(True)
        out[k] = t
    return out


# ------------------------------------------------------------------------------------------------------------------------
# A per-sequence driver built from FOREIGN torch ops, for the drop-in tests (tests/test_gpu_dropin.py).  What it stands for: a
# maintainer who swaps the import keeps an evaluator that glues the model's outputs together with
# ATen -- resampling with F.interpolate, an argmax, a one-hot of the index map, an in-place overwrite of the probability map the
# model returned, a torch.cat that grows it.  The call order it has to follow (swem_evaluator.py:59-148: key -> match -> segment
# -> [resample -> value -> memorize] on all but the last frame, late annotations injected before the argmax) is not asserted
# here: the REFERENCE's recorded index maps and logits (fixtures g6 / g7 / g8, written by the reference's own evaluator methods,
# tests/golden/make_golden.py) are the judge of it -- a wrong order or resampling mode does not reproduce them.
def aten_glue_loop(model, clip, annotations, out_hw, keep_scores=False):
    """clip (1,T,3,H,W); annotations: one (1,N+1,Ho,Wo) mask tensor or None per frame (entry 0 = the first frame's objects;
    a later entry = objects annotated for the first time at that frame, YouTube-VOS style).  Returns (index maps, scores):
    T-1 int64 maps (1,Ho,Wo) and, with keep_scores, per frame (probabilities, logits) cloned before anything overwrites them."""
    import torch.nn.functional as F
    n_frames = clip.shape[1]
    H, W = clip.shape[-2:]
    image = clip[:, 0]
    key, _, feat16, _, _ = model('encode_key', image)
    seen = annotations[0]
    value = model('encode_value', image, F.interpolate(seen, size=(H, W), mode='nearest').float(), feat16)
    model('init', key, value, seen)
    index_maps, scores = [], []
    for f in range(1, n_frames):
        image = clip[:, f]
        key, qvalue, feat16, feat8, feat4 = model('encode_key', image)
        context, n_obj = model('match', key, qvalue)
        logits, prob = model('segment', n_obj, context, feat8, feat4, None, out_hw)
        if keep_scores:
            scores.append((prob.clone(), logits.clone()))
        late = annotations[f] if f < len(annotations) else None
        if late is not None:
            fresh = late[:, 1:]                                          # the new objects' masks (channel 0 is background)
            prob.masked_fill_(fresh.sum(dim=1, keepdim=True) > 0, 0)    # in place, on the tensor the model handed out
            prob = torch.cat((prob, fresh), dim=1)
            n_obj = prob.shape[1] - 1
        idx = prob.argmax(dim=1)
        index_maps.append(idx)
        if f == n_frames - 1:
            break                                                        # the last frame is not memorized
        onehot = F.one_hot(idx, n_obj + 1).permute(0, 3, 1, 2).contiguous()           # int64, as the index map
        soft = F.interpolate(prob, size=(H, W), mode='bilinear', align_corners=False)
        model('memorize', key, model('encode_value', image, soft, feat16), onehot, soft)
    return index_maps, scores
