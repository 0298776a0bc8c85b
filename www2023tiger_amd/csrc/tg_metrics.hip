// Link-prediction metrics of the evaluation harness on device (tiger/eval_utils.py:49-67):
// for consecutive windows of `chunk` events, sklearn's average_precision_score and
// roc_auc_score of the 2*bs predictions [pos | neg] with labels [1 | 0], ties included.
//   AP  = (1/P) sum over positives i of  #pos(s >= s_i) / #all(s >= s_i)
//         (= sum over distinct thresholds of the recall step times the precision there)
//   AUC = (sum over (pos i, neg j) of [s_i > s_j] + 0.5 [s_i == s_j]) / (P N)
// Counts are integers and the ratios are formed in float64, as sklearn does.  Non-finite
// predictions are dropped from their window (eval_utils.py:57-62).
#include "tg_common.h"

namespace tg {

__global__ void __launch_bounds__(256) k_ap_auc(int64_t n, int chunk, const float* __restrict__ pos,
                                                const float* __restrict__ neg, double* __restrict__ ap,
                                                double* __restrict__ auc, int32_t* __restrict__ n_bad) {
  __shared__ double red[2][256];
  __shared__ int cnt[3];
  const int64_t lo = (int64_t)blockIdx.x * chunk;
  const int bs = (int)min((int64_t)chunk, n - lo);
  const float* p = pos + lo;
  const float* q = neg + lo;
  if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
  __syncthreads();
  int P = 0, N = 0;
  for (int i = threadIdx.x; i < bs; i += 256) {
    P += isfinite(p[i]) ? 1 : 0;
    N += isfinite(q[i]) ? 1 : 0;
  }
  atomicAdd(&cnt[0], P);
  atomicAdd(&cnt[1], N);
  __syncthreads();
  P = cnt[0];
  N = cnt[1];
  double a = 0.0, u = 0.0;
  for (int i = threadIdx.x; i < bs; i += 256) {
    const float s = p[i];
    if (!isfinite(s)) continue;
    int pge = 0, nge = 0, ngt = 0;  // positives >= s, negatives >= s, negatives > s  (finite only)
    for (int j = 0; j < bs; ++j) {
      const float x = p[j], y = q[j];
      pge += (isfinite(x) && x >= s) ? 1 : 0;
      nge += (isfinite(y) && y >= s) ? 1 : 0;
      ngt += (isfinite(y) && y > s) ? 1 : 0;
    }
    a += (double)pge / (double)(pge + nge);
    u += (double)(N - nge) + 0.5 * (double)(nge - ngt);  // negatives below s, plus half the ties
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = u;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ap[blockIdx.x] = P > 0 ? red[0][0] / (double)P : 0.0;
    auc[blockIdx.x] = (P > 0 && N > 0) ? red[1][0] / ((double)P * (double)N) : 0.0;
    if (n_bad && (P + N) != 2 * bs) atomicAdd(n_bad, 2 * bs - P - N);
  }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_ap_auc(int64_t n, int32_t chunk, const float* pos_pred, const float* neg_pred, double* ap,
                         double* auc, int32_t* n_nonfinite, void* stream) {
  if (n < 0 || chunk <= 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!pos_pred || !neg_pred || !ap || !auc) return TG_EINVAL;
  const int64_t blocks = cdiv(n, chunk);
  hipLaunchKernelGGL(k_ap_auc, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), n, (int)chunk, pos_pred, neg_pred,
                     ap, auc, n_nonfinite);
  return check_launch("tg_ap_auc");
}
