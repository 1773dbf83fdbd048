"""Why the grouped layers split every fp32 operand into THREE bf16 terms (6 matrix-core products per MAC) and not into TWO
fp16 terms (3 products): a numpy model of one K-term dot product per variant against float64, on operands shaped like the
network's (post-ReLU activations, BN-folded weights).  CPU only:  python profiles/split_precision_study.py

  fp32 chain        sequential fp32 multiply-add, one rounding per step (what an fp32 evaluation -- the oracle -- pays)
  3 x bf16          x = x0+x1+x2, w = w0+w1+w2 (exact: 3 x 8 significand bits), products x0w0, x0w1, x1w0, x0w2, x1w1, x2w0,
                    each exact in fp32, accumulated in fp32 per 16-k block in two accumulators (csrc/mlp_split.h mfma6_2acc)
  2 x fp16          x = xh+xl, w = wh+wl (2 x 11 = 22 significand bits, round to nearest; a power-of-two scale keeps the low
                    terms normal), products xh wh, xh wl, xl wh, accumulated likewise
The yardstick of tests/parity_util.py is rms(error vs float64) <= 1.0 x the fp32 chain's."""
import numpy as np

rng = np.random.default_rng(0)


def bf16_trunc_split3(a):
    a = a.astype(np.float32)
    out = []
    r = a.copy()
    for _ in range(3):
        h = (r.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)   # truncation keeps the remainder exact
        out.append(h)
        r = (r - h).astype(np.float32)
    return out


def fp16_split2(a, scale):
    a = (a.astype(np.float32) * np.float32(scale))
    h = a.astype(np.float16).astype(np.float32)
    l = (a - h).astype(np.float16).astype(np.float32)
    return h / np.float32(scale), l / np.float32(scale)


def acc_blocks(prod_groups, kblock=16):
    """fp32 accumulation as the matrix core does it: per 16-k block the products of one group are summed (modelled exactly,
    in float64, then rounded once to fp32 -- the MFMA's internal tree is wider than fp32) and added to the accumulator"""
    rows, k = prod_groups[0].shape
    accs = [np.zeros(rows, np.float32) for _ in prod_groups]
    for b0 in range(0, k, kblock):
        for i, p in enumerate(prod_groups):
            blk = p[:, b0:b0 + kblock].astype(np.float64).sum(1)
            accs[i] = (accs[i].astype(np.float64) + blk).astype(np.float32)
    total = accs[0]
    for a in accs[1:]:
        total = (total + a).astype(np.float32)
    return total


def study(K, rows=20000):
    x = np.maximum(rng.standard_normal((rows, K)) * 2.0 + 0.5, 0).astype(np.float32)        # post-ReLU activations
    w = (rng.standard_normal((rows, K)) * 0.08).astype(np.float32)                             # folded weights
    exact = (x.astype(np.float64) * w.astype(np.float64)).sum(1)
    # fp32 chain
    acc = np.zeros(rows, np.float32)
    for k in range(K):
        acc = (acc + x[:, k] * w[:, k]).astype(np.float32)        # product rounded, sum rounded (torch-CPU style, no FMA)
    e_chain = acc.astype(np.float64) - exact
    # 3 x bf16, 6 products, two accumulators (large product | five small ones)
    x0, x1, x2 = bf16_trunc_split3(x)
    w0, w1, w2 = bf16_trunc_split3(w)
    big = x0 * w0
    small = (x0.astype(np.float64) * w1 + x1.astype(np.float64) * w0 + x0.astype(np.float64) * w2 + x1.astype(np.float64) * w1 +
             x2.astype(np.float64) * w0)
    e_b3 = acc_blocks([big, small]).astype(np.float64) - exact
    # 2 x fp16, 3 products
    xh, xl = fp16_split2(x, 1.0)
    wh, wl = fp16_split2(w, 256.0)
    bigh = xh * wh
    smallh = xh.astype(np.float64) * wl + xl.astype(np.float64) * wh
    e_h2 = acc_blocks([bigh, smallh]).astype(np.float64) - exact
    # 2 x fp16, all 4 products: the small accumulator also takes xl wl
    smallh4 = smallh + xl.astype(np.float64) * wl
    e_h2_4 = acc_blocks([bigh, smallh4]).astype(np.float64) - exact
    # the same with the high term cut by truncation (v_cvt_pkrtz_f16_f32) and the low term rounded to nearest
    xa = x.astype(np.float32)
    def rtz16(a):
        h = a.astype(np.float16).astype(np.float32)
        over = np.abs(h) > np.abs(a)
        h[over] = np.nextafter(h[over].astype(np.float16), np.float16(0)).astype(np.float32)
        return h
    xht = rtz16(xa); xlt = (xa - xht).astype(np.float16).astype(np.float32)
    ws_ = (w * np.float32(256.0)).astype(np.float32)
    wht = rtz16(ws_); wlt = (ws_ - wht).astype(np.float16).astype(np.float32)
    wht, wlt = wht / np.float32(256.0), wlt / np.float32(256.0)
    big_t = xht * wht
    small_t = xht.astype(np.float64) * wlt + xlt.astype(np.float64) * wht + xlt.astype(np.float64) * wlt
    e_h2_4t = acc_blocks([big_t, small_t]).astype(np.float64) - exact
    # what the operand rounding alone costs the fp16 form (exact accumulation)
    e_h2_operands = ((xh.astype(np.float64) + xl) * (wh.astype(np.float64) + wl)).sum(1) - exact
    rms = lambda e: float(np.sqrt((e ** 2).mean()))
    r = {"K": K, "fp32_chain": rms(e_chain), "bf16x3_6prod": rms(e_b3), "fp16x2_3prod": rms(e_h2), "fp16x2_4prod": rms(e_h2_4),
         "fp16x2_4prod_rtz_hi": rms(e_h2_4t),
         "fp16x2_operand_rounding_only": rms(e_h2_operands)}
    r["bf16x3_over_chain"] = r["bf16x3_6prod"] / r["fp32_chain"]
    r["fp16x2_over_chain"] = r["fp16x2_3prod"] / r["fp32_chain"]
    return r


if __name__ == "__main__":
    import json
    out = [study(K) for K in (64, 128, 256, 512)]
    for r in out:
        print("K=%3d  rms error vs float64: fp32 chain %.3e | 3 x bf16 %.3e (%.2fx) | 2 x fp16 3 products %.3e (%.2fx), 4 products %.3e "
              "(rtz high term %.3e); operand rounding alone %.3e"
              % (r["K"], r["fp32_chain"], r["bf16x3_6prod"], r["bf16x3_over_chain"], r["fp16x2_3prod"], r["fp16x2_over_chain"],
                 r["fp16x2_4prod"], r["fp16x2_4prod_rtz_hi"], r["fp16x2_operand_rounding_only"]))
    json.dump(out, open(__file__.replace(".py", ".json"), "w"), indent=1)
