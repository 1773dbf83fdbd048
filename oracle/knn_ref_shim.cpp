// Trampoline (ours, not reference code): exposes the reference's knn_cpu()
// (knn/src/cpu/knn_cpu.cpp:4, declared in knn/src/cpu/vision.h) and the batch
// loop of knn/src/knn.h:56-62 through a C ABI so the oracle tests can call the
// reference implementation with ctypes.  Built only into oracle/_ref/.
#include <cstdlib>
void knn_cpu(float* ref_dev, int ref_width, float* query_dev, int query_width,
             int height, int k, float* dist_dev, long* ind_dev, long* ind_buf);

extern "C" int knn_ref(float* ref, float* query, long* idx, long batch, long dim,
                       long k, long ref_nb, long query_nb) {
  float* dist_dev = (float*)malloc(ref_nb * query_nb * sizeof(float));
  long* ind_buf = (long*)malloc(ref_nb * sizeof(long));
  for (long b = 0; b < batch; b++)
    knn_cpu(ref + b * dim * ref_nb, (int)ref_nb, query + b * dim * query_nb, (int)query_nb,
            (int)dim, (int)k, dist_dev, idx + b * k * query_nb, ind_buf);
  free(dist_dev);
  free(ind_buf);
  return 1;
}
