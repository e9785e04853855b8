// Which hipBLASLt epilogues exist for fp32 at the deformation MLP's GEMM shapes on this ROCm?  
// The MLP's backward (MG/resnetfc.py:10-62 ResnetBlockFC, backward of x + fc_1(relu(fc_0(relu(x))))) needs, per block and per
// data-gradient GEMM, y = (g W) * (act > 0) [+ residual] and the bias gradient colsum(y): a "dReLU (+ bias-gradient)" epilogue.
// This program asks hipblasLtMatmulAlgoGetHeuristic for fp32 / compute 32F at M = 100 000 rows, N = K = 512 with every epilogue
// the header declares, and runs the ones that return an algorithm once.  Build: make -C scripts/ubench hipblaslt_epilogues
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { auto _s = (x); if (_s != 0) { printf("  call failed: %s -> %d\n", #x, (int)_s); } } while (0)

int main() {
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  int ver = 0;
  hipblasLtGetVersion(h, &ver);
  printf("hipBLASLt version %d\n", ver);
  const int64_t M = 100000, N = 512, K = 512;
  float *A, *B, *C, *D, *bias, *aux;
  hipMalloc(&A, sizeof(float) * M * K); hipMalloc(&B, sizeof(float) * K * N); hipMalloc(&C, sizeof(float) * M * N);
  hipMalloc(&D, sizeof(float) * M * N); hipMalloc(&bias, sizeof(float) * (M > N ? M : N)); hipMalloc(&aux, sizeof(float) * M * N);
  hipMemset(A, 0, sizeof(float) * M * K); hipMemset(B, 0, sizeof(float) * K * N); hipMemset(aux, 0, sizeof(float) * M * N);
  void* ws; size_t ws_bytes = 64 << 20; hipMalloc(&ws, ws_bytes);
  struct E { const char* name; hipblasLtEpilogue_t e; bool needs_aux; };
  const std::vector<E> eps = {
      {"DEFAULT", HIPBLASLT_EPILOGUE_DEFAULT, false}, {"RELU", HIPBLASLT_EPILOGUE_RELU, false}, {"BIAS", HIPBLASLT_EPILOGUE_BIAS, false},
      {"RELU_BIAS", HIPBLASLT_EPILOGUE_RELU_BIAS, false}, {"RELU_AUX", HIPBLASLT_EPILOGUE_RELU_AUX, true},
      {"RELU_AUX_BIAS", HIPBLASLT_EPILOGUE_RELU_AUX_BIAS, true}, {"GELU_AUX_BIAS", HIPBLASLT_EPILOGUE_GELU_AUX_BIAS, true},
      {"DGELU", HIPBLASLT_EPILOGUE_DGELU, true}, {"DGELU_BGRAD", HIPBLASLT_EPILOGUE_DGELU_BGRAD, true},
      {"BGRADA", HIPBLASLT_EPILOGUE_BGRADA, false}, {"BGRADB", HIPBLASLT_EPILOGUE_BGRADB, false},
      {"CLAMP_AUX_BIAS_EXT", HIPBLASLT_EPILOGUE_CLAMP_AUX_BIAS_EXT, true}};
  printf("epilogues the header declares (hipblaslt.h hipblasLtEpilogue_t): no DRELU, no DRELU_BGRAD -- the backward-activation\n"
         "epilogues are DGELU / DGELU_BGRAD only; BGRADA / BGRADB sum an INPUT operand's columns (the bias gradient of a weight-\n"
         "gradient GEMM), they do not mask.\n");
  // column-major view of the row-major problem Y[M, N] = X[M, K] W^T: D^T (N x M) = W (N x K) X^T (K x M)
  for (const E& ep : eps) {
    hipblasLtMatmulDesc_t desc;
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep.e, sizeof(ep.e)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
    if (ep.needs_aux) {
      int64_t ld = N;
      CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_POINTER, &aux, sizeof(aux)));
      CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_LD, &ld, sizeof(ld)));
    }
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, K, N, K));   // W^T stored as W [N, K] row-major = [K, N] column-major, op T
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, K, M, K));   // X [M, K] row-major = [K, M] column-major
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, N, M, N));
    hipblasLtMatmulPreference_t pref;
    CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
    hipblasLtMatmulHeuristicResult_t res[8];
    int n = 0;
    hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 8, res, &n);
    printf("%-20s fp32 M=%ld N=%ld K=%ld: heuristic status %d, %d algorithm(s)", ep.name, (long)M, (long)N, (long)K, (int)st, n);
    if (st == HIPBLAS_STATUS_SUCCESS && n > 0) {
      const float one = 1.f, zero = 0.f;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipblasStatus_t rs = HIPBLAS_STATUS_SUCCESS;
      for (int it = 0; it < 6 && rs == HIPBLAS_STATUS_SUCCESS; it++) {
        if (it == 1) hipEventRecord(e0, 0);
        rs = hipblasLtMatmul(h, desc, &one, B, la, A, lb, &zero, C, lc, D, lc, &res[0].algo, ws, ws_bytes, 0);
      }
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      printf(", run status %d, %.3f ms per GEMM = %.1f TFLOP/s", (int)rs, ms / 5, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12);
    }
    printf("\n");
    hipblasLtMatmulPreferenceDestroy(pref); hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb);
    hipblasLtMatrixLayoutDestroy(lc); hipblasLtMatmulDescDestroy(desc);
  }
  hipblasLtDestroy(h);
  return 0;
}
