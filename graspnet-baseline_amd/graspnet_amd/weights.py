"""Host-side weight preparation: fold eval-mode BatchNorm into the 1x1 convs and pack the result in
the MFMA fragment order the kernels read (csrc/mlp_core.h).

Input is a state-dict with the reference's key layout (models/graspnet.py:63-68 and children; see
tests/golden/state_dict_manifest.txt), e.g. what `torch.load(ckpt)['model_state_dict']` holds.
"""
import torch

BN_EPS = 1e-5   # torch.nn.BatchNorm1d/2d default used by pytorch_utils.py:40-60, modules.py:38-39


class Folded(tuple):
    """(W', b') float32 as a plain 2-tuple, plus .w64: the folded weight before its rounding to float32.  The split-bf16
    fragments are cut from .w64 (three bf16 terms hold ~27 significant bits, so the fold costs them no rounding)."""
    def __new__(cls, w64, b64):
        self = super().__new__(cls, (w64.float(), b64.float()))
        self.w64 = w64
        return self

    def cols(self, lo, hi=None):
        """the same layer restricted to input columns lo:hi (bias kept)"""
        return Folded(self.w64[:, lo:hi], self[1].double())


def fold_bn(w, b, sd, bn):
    """conv (Cout,Cin[,1[,1]]) [+bias] followed by BN(eval) -> Folded (W', b'), computed in float64."""
    w = w.reshape(w.shape[0], -1).double()
    cout = w.shape[0]
    b = torch.zeros(cout, dtype=torch.float64) if b is None else b.double()
    if bn is not None:
        scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS)
        w = w * scale[:, None]
        b = (b - sd[bn + ".running_mean"].double()) * scale + sd[bn + ".bias"].double()
    return Folded(w, b)


def pad_to(n, m):
    return (n + m - 1) // m * m


def coutp_for(cout):
    """COUT padded so its 32-wide tiles divide over the 4 waves (csrc/mlp_core.h wn_for)."""
    p = pad_to(cout, 32)
    return pad_to(p, 128) if p // 32 >= 4 else p


def pack_fragments(w, b, cinp=None, coutp=None):
    """(Cout,Cin) f32, (Cout,) -> Wp[nt][kb][lane][s] = W[nt*32+(lane&31)][kb*8+4*(lane>>5)+s], padded bias."""
    cout, cin = w.shape
    cinp = pad_to(cin, 8) if cinp is None else cinp
    coutp = coutp_for(cout) if coutp is None else coutp
    wp = torch.zeros(coutp, cinp, dtype=torch.float32)
    wp[:cout, :cin] = w
    bp = torch.zeros(coutp, dtype=torch.float32)
    bp[:cout] = b
    nt, kb = coutp // 32, cinp // 8
    frag = wp.reshape(nt, 32, kb, 2, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(nt, kb, 64, 4)
    return frag.contiguous(), bp


def split3(w):
    """fp32 or fp64 -> three bf16 terms with w == hi + mid + lo to ~2^-27 relative (each bf16 rounding leaves a
    residual <= 2^-9 of the term before it; the residuals are exact in the input's own precision)."""
    hi = w.float().to(torch.bfloat16) if w.dtype == torch.float64 else w.to(torch.bfloat16)
    r = w - hi.to(w.dtype)
    mid = r.float().to(torch.bfloat16)
    lo = (r - mid.to(w.dtype)).float().to(torch.bfloat16)
    return hi, mid, lo


def pack_split(w, b, cinp=None, coutp=None):
    """(Cout,Cin) f32 or f64 -> Wq[nt][kb][plane][lane][j] bf16 = W_plane[nt*32+(lane&31)][kb*16+8*(lane>>5)+j] (csrc/mlp_split.h),
    padded fp32 bias."""
    cout, cin = w.shape
    cinp = pad_to(cin, 16) if cinp is None else cinp
    coutp = coutp_for(cout) if coutp is None else coutp
    wp = torch.zeros(coutp, cinp, dtype=w.dtype)
    wp[:cout, :cin] = w
    bp = torch.zeros(coutp, dtype=torch.float32)
    bp[:cout] = b
    nt, kb = coutp // 32, cinp // 16
    planes = torch.stack(split3(wp), 0)                                   # (3, coutp, cinp)
    frag = planes.reshape(3, nt, 32, kb, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous().reshape(nt, kb, 3, 64, 8)
    return frag.contiguous(), bp


class PackedMlp:
    """Device-resident packed layers + the ctypes pointer arrays the C ABI takes."""

    def __init__(self, layers, device):
        import ctypes
        self.w = [w.to(device) for w, _ in layers]
        self.b = [b.to(device) for _, b in layers]
        n = len(layers)
        self.wptr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in self.w])
        self.bptr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in self.b])


def _shared_mlp_layers(sd, prefix, nlayers):
    return [fold_bn(sd["%s.layer%d.conv.weight" % (prefix, i)], None, sd, "%s.layer%d.bn.bn" % (prefix, i))
            for i in range(nlayers)]


def prepare(sd, device, num_view=300, num_angle=12):
    """-> dict of PackedMlp for every stage."""
    sd = {k: v.detach().cpu() for k, v in sd.items()}
    out = {}
    bb = "view_estimator.backbone"
    for li in range(4):
        layers = _shared_mlp_layers(sd, "%s.sa%d.mlp_module" % (bb, li + 1), 3)
        w0, b0 = layers[0]
        c = w0.shape[1] - 3
        # reference input order is [xyz(3) | features(C)] (pointnet2_utils.py:353); kernels use [features | xyz | pad]
        w0 = torch.cat([w0[:, 3:], w0[:, :3]], dim=1)
        w0_64 = torch.cat([layers[0].w64[:, 3:], layers[0].w64[:, :3]], dim=1)
        tail_split = [pack_split(l.w64, l[1]) for l in layers[1:]]
        packed = [pack_fragments(w0, b0, cinp=c + 8)] + [pack_fragments(w, b) for w, b in layers[1:]]
        out["sa%d" % (li + 1)] = PackedMlp(packed, device)
        if c == 0:   # SA1 on the matrix-core path: layer 1 (K = 3) fp32, layers 2-3 split
            out["sa1_split"] = PackedMlp([pack_fragments(w0, b0, cinp=8)] + tail_split, device)
        if c > 0:
            # layer 1 split for the hoisted kernel: per-point part W1f (+ bias) and per-neighbour K=8 block W1x
            out["sa%d_pre" % (li + 1)] = PackedMlp([pack_fragments(w0[:, :c], b0)], device)
            out["sa%d_pre_split" % (li + 1)] = PackedMlp([pack_split(w0_64[:, :c], b0)], device)
            out["sa%d_hoist" % (li + 1)] = PackedMlp(
                [pack_fragments(w0[:, c:], torch.zeros_like(b0), cinp=8)] + [pack_fragments(w, b) for w, b in layers[1:]],
                device)
            out["sa%d_hoist_split" % (li + 1)] = PackedMlp(
                [pack_fragments(w0[:, c:], torch.zeros_like(b0), cinp=8)] + tail_split, device)
    for i in (1, 2):
        layers = _shared_mlp_layers(sd, "%s.fp%d.mlp" % (bb, i), 2)
        out["fp%d" % i] = PackedMlp([pack_fragments(w, b) for w, b in layers], device)
        out["fp%d_split" % i] = PackedMlp([pack_split(l.w64, l[1]) for l in layers], device)
    vp = "view_estimator.vpmodule"
    nv = 2 + num_view
    l1 = fold_bn(sd[vp + ".conv1.weight"], sd[vp + ".conv1.bias"], sd, vp + ".bn1")
    l2 = fold_bn(sd[vp + ".conv2.weight"], sd[vp + ".conv2.bias"], sd, vp + ".bn2")
    l3 = fold_bn(sd[vp + ".conv3.weight"], sd[vp + ".conv3.bias"], sd, None)
    # one kernel shape for every num_view <= 318: 256 -> 256 -> 384 -> 384, layer 3's K padded to 320 (zeros beyond 2+V)
    if not 3 <= nv <= 320:
        raise NotImplementedError("num_view must be 1..318 (got %d)" % num_view)
    nvp, k3 = 384, 320
    out["approach"] = PackedMlp([pack_fragments(*l1), pack_fragments(*l2, coutp=nvp),
                                 pack_fragments(*l3, cinp=k3, coutp=nvp)], device)
    crop_layers = _shared_mlp_layers(sd, "grasp_generator.crop.mlps", 3)
    out["approach_split"] = PackedMlp([pack_split(l1.w64, l1[1]), pack_split(l2.w64, l2[1], coutp=nvp),
                                       pack_split(l3.w64, l3[1], cinp=k3, coutp=nvp)], device)
    out["crop"] = PackedMlp([pack_fragments(w, b, cinp=8 if i == 0 else None) for i, (w, b) in enumerate(crop_layers)], device)
    # layers 2-3 as split-bf16 fragments for the matrix-core path (layer 1, K = 3, stays fp32)
    out["crop_split"] = PackedMlp([pack_fragments(*crop_layers[0], cinp=8), pack_split(crop_layers[1].w64, crop_layers[1][1]),
                                   pack_split(crop_layers[2].w64, crop_layers[2][1])], device)
    # OperationNet + ToleranceNet as one chain: conv1 stacked, conv2 / conv3 block-diagonal
    op, tl = "grasp_generator.operation", "grasp_generator.tolerance"
    o1 = fold_bn(sd[op + ".conv1.weight"], sd[op + ".conv1.bias"], sd, op + ".bn1")
    t1 = fold_bn(sd[tl + ".conv1.weight"], sd[tl + ".conv1.bias"], sd, tl + ".bn1")
    o2 = fold_bn(sd[op + ".conv2.weight"], sd[op + ".conv2.bias"], sd, op + ".bn2")
    t2 = fold_bn(sd[tl + ".conv2.weight"], sd[tl + ".conv2.bias"], sd, tl + ".bn2")
    o3 = fold_bn(sd[op + ".conv3.weight"], sd[op + ".conv3.bias"], sd, None)
    t3 = fold_bn(sd[tl + ".conv3.weight"], sd[tl + ".conv3.bias"], sd, None)
    h = o1[0].shape[0]                       # 128
    w1 = torch.cat([o1[0], t1[0]], 0); b1 = torch.cat([o1[1], t1[1]], 0)
    w2 = torch.zeros(2 * h, 2 * h); w2[:h, :h] = o2[0]; w2[h:, h:] = t2[0]
    b2 = torch.cat([o2[1], t2[1]], 0)
    no, nt_ = o3[0].shape[0], t3[0].shape[0]   # 36, 12
    w3 = torch.zeros(no + nt_, 2 * h); w3[:no, :h] = o3[0]; w3[no:, h:] = t3[0]
    b3 = torch.cat([o3[1], t3[1]], 0)
    out["heads"] = PackedMlp([pack_fragments(w1, b1), pack_fragments(w2, b2), pack_fragments(w3, b3)], device)
    w1d = torch.cat([o1.w64, t1.w64], 0)
    w2d = torch.zeros(2 * h, 2 * h, dtype=torch.float64); w2d[:h, :h] = o2.w64; w2d[h:, h:] = t2.w64
    w3d = torch.zeros(no + nt_, 2 * h, dtype=torch.float64); w3d[:no, :h] = o3.w64; w3d[no:, h:] = t3.w64
    out["heads_split"] = PackedMlp([pack_split(w1d, b1), pack_split(w2d, b2), pack_split(w3d, b3)], device)
    return out
