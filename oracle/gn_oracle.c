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
 * Evaluation policy of the index-deciding float expressions (gn_oracle_fp.inc): 0 as written (default), 1 / 2 contracted
 * into fused multiply-adds in the two plausible orders.  Same switch as the HIP library's gn_set_fp_policy().
 */
static int g_fp_policy = 0;
void gno_set_fp_policy(int p) { g_fp_policy = (p == 1 || p == 2) ? p : 0; }
int gno_get_fp_policy(void) { return g_fp_policy; }

#define GNO_FP 0
#include "gn_oracle_fp.inc"
#undef GNO_FP
#define GNO_FP 1
#include "gn_oracle_fp.inc"
#undef GNO_FP
#define GNO_FP 2
#include "gn_oracle_fp.inc"
#undef GNO_FP

#define GNO_DISPATCH(fn, ...)                                   \
  do {                                                          \
    if (g_fp_policy == 1) fn##_fp1(__VA_ARGS__);                \
    else if (g_fp_policy == 2) fn##_fp2(__VA_ARGS__);           \
    else fn##_fp0(__VA_ARGS__);                                 \
  } while (0)

/* sampling_gpu.cu:74-178 + sampling.cpp:70-91 (body: gn_oracle_fp.inc) */
void gno_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int32_t *idxs) {
  GNO_DISPATCH(gno_furthest_point_sampling, b, n, m, dataset, temp, idxs);
}
/* ball_query_gpu.cu:14-49 */
void gno_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int32_t *idx) {
  GNO_DISPATCH(gno_ball_query, b, n, m, radius, nsample, new_xyz, xyz, idx);
}
/* cylinder_query_gpu.cu:9-57 */
void gno_cylinder_query(int b, int n, int m, float radius, float hmin, float hmax, int nsample, const float *new_xyz,
                        const float *xyz, const float *rot, int32_t *idx) {
  GNO_DISPATCH(gno_cylinder_query, b, n, m, radius, hmin, hmax, nsample, new_xyz, xyz, rot, idx);
}
/* interpolate_gpu.cu:14-64 */
void gno_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx) {
  GNO_DISPATCH(gno_three_nn, b, n, m, unknown, known, dist2, idx);
}
/* knn.h:12-65, knn_cpu.cpp:4-56, knn.cu:36-176 */
void gno_knn(int b, int dim, int nr, int nq, int k, const float *ref, const float *query, int64_t *idx) {
  GNO_DISPATCH(gno_knn, b, dim, nr, nq, k, ref, query, idx);
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
