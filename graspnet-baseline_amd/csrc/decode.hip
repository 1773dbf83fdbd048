// decode.hip -- view selection after ApproachNet and grasp decoding, each as one kernel.
//
//   approach_post : models/modules.py:64-76  (top view arg-max over 300 scores, template lookup,
//                   view -> rotation matrix).  The reference rebuilds the 300-view template on the
//                   HOST every forward and copies it to the device (modules.py:66,
//                   loss_utils.py:59-66); here the template and the 300 view rotation matrices are
//                   device tables built once at model load (graspnet_amd/geometry.py).
//   pred_decode   : models/graspnet.py:77-134 (python loop over the batch, ~25 tiny kernels per
//                   scene) -> one workgroup per scene writing padded (M,17) rows + a count.
//                   The in-plane rotation R(view) * Rx(angle) takes only 300*12 values, so it is a
//                   table lookup too (built on the host with the reference's formula).
// Arg-max ties follow torch: first maximal index.
#include "gn_common.h"

namespace gn {

// one wave per seed row.  feat: (rows, ldf) = [objectness(2) | view scores(nview)]
__global__ __launch_bounds__(256) void approach_post_kernel(const float *__restrict__ feat, int ldf, int rows,
                                                            int nview, const float *__restrict__ views,
                                                            const float *__restrict__ view_rot,
                                                            int64_t *__restrict__ top_inds,
                                                            float *__restrict__ top_score,
                                                            float *__restrict__ top_xyz,
                                                            float *__restrict__ top_rot) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *sc = feat + (size_t)row * ldf + 2;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = lane; v < nview; v += 64) {
    const float s = sc[v];
    if (s > best || bi == 0x7fffffff) { best = s; bi = v; }
  }
  const float wmax = wave_max(best);
  const int wi = wave_min_i((best == wmax && bi != 0x7fffffff) ? bi : 0x7fffffff);
  const int ind = wi == 0x7fffffff ? 0 : wi;
  if (lane == 0) {
    top_inds[row] = ind;
    top_score[row] = wmax;
  }
  if (lane < 3) top_xyz[(size_t)row * 3 + lane] = views[ind * 3 + lane];
  if (lane < 9) top_rot[(size_t)row * 9 + lane] = view_rot[ind * 9 + lane];
}

// one workgroup per scene
__global__ __launch_bounds__(1024) void pred_decode_kernel(const float *__restrict__ feat, int ldf,
                                                           const float *__restrict__ head,
                                                           const int64_t *__restrict__ top_inds,
                                                           const float *__restrict__ seed_xyz,
                                                           const float *__restrict__ rot_table, int m, int nd,
                                                           int na, int vec16, float *__restrict__ grasps,
                                                           int32_t *__restrict__ counts) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hc = 4 * na;  // channels per (seed, depth): score | angle cls | width | tolerance
  feat += (size_t)b * m * ldf;
  head += (size_t)b * m * nd * hc;
  top_inds += (size_t)b * m;
  seed_xyz += (size_t)b * m * 3;
  grasps += (size_t)b * m * 17;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int s0 = 0; s0 < m; s0 += 1024) {
    const int s = s0 + tid;
    bool keep = false;
    float score = 0.f, width = 0.f, tol = 0.f;
    int bd = 0, ba = 0;
    if (s < m) {
      const float *f = feat + (size_t)s * ldf;
      keep = f[1] > f[0];  // argmax(objectness,0)==1  (graspnet.py:112-113)
      if (na == 12 && vec16 && nd == 4) {
        // the reference configuration: a (seed, depth) row is 48 floats = [score | angle cls | width | tolerance] x 12.  Two
        // round trips for the whole seed instead of one per depth: the four depths' angle-class quarters (3 x 16 bytes each,
        // all 12 loads in flight), then the three values each depth's arg-max picks (12 dword loads in flight) -- the walk
        // that loaded every depth's 12 x 16 bytes in turn made this kernel 28 us, a chain of ~50 dependent dwords before it 45
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        const float *hs = head + (size_t)s * 4 * hc;
        f32x4 q[4][3];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int i = 0; i < 3; ++i) q[d][i] = reinterpret_cast<const f32x4 *>(hs + d * hc + 12)[i];
        int a4[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          int a = 0;
          float av = q[d][0][0];
#pragma unroll
          for (int t = 1; t < 12; ++t) {  // graspnet.py:93 angle class per (seed, depth): first maximum
            const float v = q[d][t >> 2][t & 3];
            if (v > av) { av = v; a = t; }
          }
          a4[d] = a;
        }
        float sc4[4], wd4[4], tl4[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          sc4[d] = hs[d * hc + a4[d]];
          wd4[d] = hs[d * hc + 24 + a4[d]];
          tl4[d] = hs[d * hc + 36 + a4[d]];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
          if (d == 0 || sc4[d] > score) {  // graspnet.py:103 depth class = argmax of the gathered score
            score = sc4[d]; bd = d; ba = a4[d]; width = wd4[d]; tol = tl4[d];
          }
      } else if (na == 12 && vec16) {
        // the reference configuration: a (seed, depth) row is 48 floats = 12 x 16-byte loads, all requested before the
        // first compare (the scalar walk below chains ~50 dependent 4-byte loads per depth and made this kernel 45 us)
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        for (int d = 0; d < nd; ++d) {
          const f32x4 *h4 = reinterpret_cast<const f32x4 *>(head + ((size_t)s * nd + d) * hc);
          f32x4 q[12];
#pragma unroll
          for (int i = 0; i < 12; ++i) q[i] = h4[i];
          int a = 0;
          float av = q[3][0];
#pragma unroll
          for (int t = 1; t < 12; ++t) {  // graspnet.py:93 angle class per (seed, depth): first maximum
            const float v = q[3 + (t >> 2)][t & 3];
            if (v > av) { av = v; a = t; }
          }
          float sc = 0.f, wd = 0.f, tl = 0.f;
#pragma unroll
          for (int t = 0; t < 12; ++t) {
            if (t == a) { sc = q[t >> 2][t & 3]; wd = q[6 + (t >> 2)][t & 3]; tl = q[9 + (t >> 2)][t & 3]; }
          }
          if (d == 0 || sc > score) {  // graspnet.py:103 depth class = argmax of the gathered score
            score = sc; bd = d; ba = a; width = wd; tol = tl;
          }
        }
      } else {
      for (int d = 0; d < nd; ++d) {
        const float *h = head + ((size_t)s * nd + d) * hc;
        int a = 0;
        float av = h[na];
        for (int t = 1; t < na; ++t) {  // graspnet.py:93 angle class per (seed, depth)
          const float v = h[na + t];
          if (v > av) { av = v; a = t; }
        }
        const float sc = h[a];
        if (d == 0 || sc > score) {  // graspnet.py:103 depth class = argmax of the gathered score
          score = sc; bd = d; ba = a;
          width = h[2 * na + a];
          tol = h[3 * na + a];
        }
      }
      }
    }
    const unsigned long long mask = __ballot(keep);
    const int wprefix = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (lane == 0) s_wave[wave] = __popcll(mask);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (keep) {
      float *g = grasps + (size_t)(off + wprefix) * 17;
      float w12 = 1.2f * width;                      // graspnet.py:87
      w12 = fminf(fmaxf(w12, 0.f), 0.1f);            // clamp(0, GRASP_MAX_WIDTH)
      g[0] = score * tol / 0.05f;                    // graspnet.py:121 (GRASP_MAX_TOLERANCE)
      g[1] = w12;
      g[2] = 0.02f;                                  // graspnet.py:131
      g[3] = ((float)bd + 1.0f) * 0.01f;             // graspnet.py:104
      const float *r = rot_table + ((size_t)top_inds[s] * na + ba) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) g[4 + i] = r[i];
      g[13] = seed_xyz[s * 3 + 0];
      g[14] = seed_xyz[s * 3 + 1];
      g[15] = seed_xyz[s * 3 + 2];
      g[16] = -1.0f;                                 // obj_ids
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  const int total = s_base;
  if (tid == 0) counts[b] = total;
  for (int e = total * 17 + tid; e < m * 17; e += 1024) grasps[e] = 0.f;  // padding rows
}

}  // namespace gn

using namespace gn;

extern "C" int gn_approach_post(const float *feat, int ldf, int rows, int nview, const float *views,
                                const float *view_rot, int64_t *top_inds, float *top_score, float *top_xyz,
                                float *top_rot, gn_stream_t stream) {
  GN_REQUIRE(rows >= 0 && nview >= 1 && ldf >= nview + 2, "approach_post: bad sizes");
  if (rows == 0) return GN_OK;
  GN_REQUIRE(feat && views && view_rot && top_inds && top_score && top_xyz && top_rot, "approach_post: null pointer");
  hipLaunchKernelGGL(approach_post_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), feat, ldf, rows,
                     nview, views, view_rot, top_inds, top_score, top_xyz, top_rot);
  GN_CHECK_LAUNCH("approach_post");
  return GN_OK;
}

extern "C" int gn_pred_decode(const float *feat, int ldf, const float *head, const int64_t *top_inds,
                              const float *seed_xyz, const float *rot_table, int b, int m, int ndepth, int nangle,
                              float *grasps, int32_t *counts, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && m >= 0 && ndepth >= 1 && nangle >= 1 && ldf >= 2, "pred_decode: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(feat && head && top_inds && seed_xyz && rot_table && grasps && counts, "pred_decode: null pointer");
  const int vec16 = (reinterpret_cast<uintptr_t>(head) & 15) == 0 ? 1 : 0;     // 16-byte row loads need an aligned tensor
  hipLaunchKernelGGL(pred_decode_kernel, dim3(b), dim3(1024), 0, as_stream(stream), feat, ldf, head, top_inds,
                     seed_xyz, rot_table, m, ndepth, nangle, vec16, grasps, counts);
  GN_CHECK_LAUNCH("pred_decode");
  return GN_OK;
}
