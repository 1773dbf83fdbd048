/*
 * gn_oracle.c -- CPU ORACLE for the GraspNet point-set operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / reported CPU baseline.
 *
 * Each function is a scalar restatement of one reference CUDA kernel
 * (paths relative to the reference checkout), evaluating every floating-point
 * expression exactly as written there: left-to-right, one rounding per
 * operation (compile with -ffp-contract=off; no FMA contraction).  Threads of
 * a reference block are emulated sequentially wherever their only interaction
 * is a __syncthreads()-separated reduction, so tie orders are the reference's.
 *
 * Parity status: the reference holds NO tests / golden vectors for these
 * kernels, and its CUDA sources cannot be compiled here
 * (no nvcc).  The oracle is pinned by (a) the reference's own knn CPU source
 * built into oracle/_ref (bit-exact index comparison), and (b) the reference
 * python model imported on top of this library to produce tests/golden/.
 * For the seven pointnet2 kernels themselves parity is otherwise UNPINNED.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* pointnet2/_ext_src/include/cuda_utils.h:18-24  opt_n_threads() */
static int ref_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

int gno_opt_n_threads(int work_size) { return ref_opt_n_threads(work_size); }

/*
 * pointnet2/_ext_src/src/sampling_gpu.cu:74-178 furthest_point_sampling_kernel
 * and host side sampling.cpp:70-91 (temp filled with 1e10, idx zero-initialised).
 *   dataset (b,n,3) f32 ; temp (b,n) f32 scratch ; idxs (b,m) i32
 * Block size = opt_n_threads(n); thread tid visits k = tid, tid+bs, ...;
 * per-thread best uses strict '>' (:113-114); tree reduction __update()
 * (:64-70) keeps the LOWER slot on ties.
 */
void gno_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                 float *temp, int32_t *idxs) {
  if (m <= 0) return;
  const int bs = ref_opt_n_threads(n);
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *ds = dataset + (size_t)bi * n * 3;
    float *tmp = temp + (size_t)bi * n;
    int32_t *out = idxs + (size_t)bi * m;
    float *dists = (float *)malloc(sizeof(float) * bs);
    int *dists_i = (int *)malloc(sizeof(int) * bs);
    for (int k = 0; k < n; ++k) tmp[k] = 1e10f; /* sampling.cpp:78-80 */
    for (int j = 0; j < m; ++j) out[j] = 0;     /* torch::zeros */
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = ds[old * 3 + 0];
      const float y1 = ds[old * 3 + 1];
      const float z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += bs) {
          const float x2 = ds[k * 3 + 0];
          const float y2 = ds[k * 3 + 1];
          const float z2 = ds[k * 3 + 2];
          const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
          /* :106 compares the float against the double literal 1e-3 */
          if ((double)mag <= 1e-3) continue;
          const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                          (z2 - z1) * (z2 - z1);
          const float d2 = fminf(d, tmp[k]);
          tmp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int half = bs / 2; half >= 1; half >>= 1) { /* :120-173 */
        for (int tid = 0; tid < half; ++tid) {
          const float v1 = dists[tid], v2 = dists[tid + half];
          const int i1 = dists_i[tid], i2 = dists_i[tid + half];
          dists[tid] = fmaxf(v1, v2);
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
    free(dists);
    free(dists_i);
  }
}

/* sampling_gpu.cu:13-25 gather_points_kernel: out[b,c,j] = points[b,c,idx[b,j]] */
void gno_gather_points(int b, int c, int n, int m, const float *points,
                       const int32_t *idx, float *out) {
#pragma omp parallel for collapse(2)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[(size_t)i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* sampling_gpu.cu:39-52 gather_points_grad_kernel (atomicAdd scatter; here in
 * ascending j order, which is one of the orders the atomics may take). */
void gno_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                            const int32_t *idx, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[(size_t)i * m + j];
        grad_points[((size_t)i * c + l) * n + a] +=
            grad_out[((size_t)i * c + l) * m + j];
      }
}

/*
 * ball_query_gpu.cu:14-49 query_ball_point_kernel; idx pre-zeroed by
 * ball_query.cpp:24-26.  new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample).
 */
void gno_ball_query(int b, int n, int m, float radius, int nsample,
                    const float *new_xyz, const float *xyz, int32_t *idx) {
  const float radius2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *px = xyz + (size_t)bi * n * 3;
      const float *pc = new_xyz + (size_t)bi * m * 3;
      int32_t *row = idx + ((size_t)bi * m + j) * nsample;
      for (int l = 0; l < nsample; ++l) row[l] = 0;
      const float new_x = pc[j * 3 + 0];
      const float new_y = pc[j * 3 + 1];
      const float new_z = pc[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = px[k * 3 + 0];
        const float y = px[k * 3 + 1];
        const float z = px[k * 3 + 2];
        const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) +
                         (new_z - z) * (new_z - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) row[l] = k;
          row[cnt] = k;
          ++cnt;
        }
      }
    }
}

/*
 * cylinder_query_gpu.cu:9-57 query_cylinder_point_kernel.
 * rot (b,m,9) row-major; the test uses (p-c)^T R (:39-44).
 */
void gno_cylinder_query(int b, int n, int m, float radius, float hmin,
                        float hmax, int nsample, const float *new_xyz,
                        const float *xyz, const float *rot, int32_t *idx) {
  const float radius2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *px = xyz + (size_t)bi * n * 3;
      const float *pc = new_xyz + (size_t)bi * m * 3;
      const float *pr = rot + ((size_t)bi * m + j) * 9;
      int32_t *row = idx + ((size_t)bi * m + j) * nsample;
      for (int l = 0; l < nsample; ++l) row[l] = 0;
      const float new_x = pc[j * 3 + 0];
      const float new_y = pc[j * 3 + 1];
      const float new_z = pc[j * 3 + 2];
      const float r0 = pr[0], r1 = pr[1], r2 = pr[2];
      const float r3 = pr[3], r4 = pr[4], r5 = pr[5];
      const float r6 = pr[6], r7 = pr[7], r8 = pr[8];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = px[k * 3 + 0] - new_x;
        const float y = px[k * 3 + 1] - new_y;
        const float z = px[k * 3 + 2] - new_z;
        const float x_rot = r0 * x + r3 * y + r6 * z;
        const float y_rot = r1 * x + r4 * y + r7 * z;
        const float z_rot = r2 * x + r5 * y + r8 * z;
        const float d2 = y_rot * y_rot + z_rot * z_rot;
        if (d2 < radius2 && x_rot > hmin && x_rot < hmax) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) row[l] = k;
          row[cnt] = k;
          ++cnt;
        }
      }
    }
}

/* group_points_gpu.cu:13-33: out[b,c,j,k] = points[b,c,idx[b,j,k]] */
void gno_group_points(int b, int c, int n, int npoints, int nsample,
                      const float *points, const int32_t *idx, float *out) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * n;
      const int32_t *ix = idx + (size_t)bi * npoints * nsample;
      float *o = out + ((size_t)bi * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          o[(size_t)j * nsample + k] = p[ix[(size_t)j * nsample + k]];
    }
}

/* group_points_gpu.cu:48-69 group_points_grad_kernel (scatter-add, j,k order) */
void gno_group_points_grad(int b, int c, int n, int npoints, int nsample,
                           const float *grad_out, const int32_t *idx,
                           float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *g = grad_points + ((size_t)bi * c + l) * n;
      const int32_t *ix = idx + (size_t)bi * npoints * nsample;
      const float *go = grad_out + ((size_t)bi * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          g[ix[(size_t)j * nsample + k]] += go[(size_t)j * nsample + k];
    }
}

/*
 * interpolate_gpu.cu:14-64 three_nn_kernel.  unknown (b,n,3), known (b,m,3)
 * -> dist2 (b,n,3) SQUARED, idx (b,n,3).  bests are doubles initialised to
 * 1e40 while d is float (:33,:39-54); strict '<' cascade, ascending k.
 */
void gno_three_nn(int b, int n, int m, const float *unknown, const float *known,
                  float *dist2, int32_t *idx) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < n; ++j) {
      const float *u = unknown + ((size_t)bi * n + j) * 3;
      const float *kn = known + (size_t)bi * m * 3;
      const float ux = u[0], uy = u[1], uz = u[2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0];
        const float y = kn[k * 3 + 1];
        const float z = kn[k * 3 + 2];
        const float d =
            (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d;     besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d;     besti2 = k;
        } else if (d < best3) {
          best3 = d;     besti3 = k;
        }
      }
      float *od = dist2 + ((size_t)bi * n + j) * 3;
      int32_t *oi = idx + ((size_t)bi * n + j) * 3;
      od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
}

/* interpolate_gpu.cu:77-106 three_interpolate_kernel */
void gno_three_interpolate(int b, int c, int m, int n, const float *points,
                           const int32_t *idx, const float *weight,
                           float *out) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * m;
      const int32_t *ix = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      float *o = out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = ix[j * 3 + 0], i2 = ix[j * 3 + 1], i3 = ix[j * 3 + 2];
        o[j] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
      }
    }
}

/* interpolate_gpu.cu:121-148 three_interpolate_grad_kernel (scatter-add) */
void gno_three_interpolate_grad(int b, int c, int n, int m,
                                const float *grad_out, const int32_t *idx,
                                const float *weight, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *g = grad_points + ((size_t)bi * c + l) * m;
      const int32_t *ix = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      const float *go = grad_out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        g[ix[j * 3 + 0]] += go[j] * w[j * 3 + 0];
        g[ix[j * 3 + 1]] += go[j] * w[j * 3 + 1];
        g[ix[j * 3 + 2]] += go[j] * w[j * 3 + 2];
      }
    }
}

/*
 * knn/src/cpu/knn_cpu.cpp:4-56 knn_cpu + knn/src/knn.h:12-65 batch loop.
 * ref (b,dim,nr), query (b,dim,nq) -> idx (b,k,nq) int64, 1-BASED.
 * The reference bubble-sorts the whole distance column (stable: swaps only on
 * strict '>'); the k smallest of a stable sort are restated as a stable
 * selection: distance ascending, lower ref index first among equal distances.
 * Distance accumulation order over dim is the reference's (:12-15), which is
 * also knn.cu:36-101's order (zero padding adds +0).
 */
void gno_knn(int b, int dim, int nr, int nq, int k, const float *ref,
             const float *query, int64_t *idx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int q = 0; q < nq; ++q) {
      const float *r = ref + (size_t)bi * dim * nr;
      const float *qq = query + (size_t)bi * dim * nq;
      int64_t *out = idx + (size_t)bi * k * nq;
      float *bd = (float *)malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
      int64_t *bi_ = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
      int cnt = 0;
      for (int ri = 0; ri < nr; ++ri) {
        float d = 0;
        for (int hi = 0; hi < dim; ++hi)
          d += (r[(size_t)hi * nr + ri] - qq[(size_t)hi * nq + q]) *
               (r[(size_t)hi * nr + ri] - qq[(size_t)hi * nq + q]);
        /* stable insertion: goes after every element with dist <= d */
        int pos = cnt;
        while (pos > 0 && bd[pos - 1] > d) --pos;
        if (pos >= k) continue;
        const int last = cnt < k ? cnt : k - 1;
        for (int t = last; t > pos; --t) { bd[t] = bd[t - 1]; bi_[t] = bi_[t - 1]; }
        bd[pos] = d;
        bi_[pos] = ri + 1;
        if (cnt < k) ++cnt;
      }
      for (int i = 0; i < k; ++i) out[(size_t)i * nq + q] = i < cnt ? bi_[i] : 0;
      free(bd);
      free(bi_);
    }
}

int gno_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void gno_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
