// EXPERIMENTS BUILD ONLY (make EXPERIMENTS=1; textually included by ../gemm_nt.hip): the launcher of the dy^T x form on the LDS-DMA pipeline with
// a stream-K tail (RE2E_TN2, kernel MODE 1) -- built twice in round 5, measured, not shipped (profiles/r05_gemm_tn_variants*.txt).  The shipped
// library answers "not taken" for it.
size_t gemm_tn2_workspace_bytes(int M, int N, int K) {
  if (!exp_env("RE2E_TN2")) return 0;
  if (M % 4 || N % 4) return 0;
  const size_t a = nt2_plan(M, N, K, false, true).bytes, b = nt2_plan(M, N, K, true, true).bytes;
  return a > b ? a : b;
}

// C[M,N] = A[K,M]^T B[K,N] (+ beta C): the weight-gradient form.  Returns 1 when launched here, 0 when left to igemm.hip.
int gemm_tn2(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, const float* bias, const float* bias2,
             int act, float beta, void* ws, size_t wsb, hipStream_t st) {
  if (!exp_env("RE2E_TN2")) return 0;
  if (M % 4 || N % 4 || lda % 4 || ldb % 4 || ldc % 4 || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) ||
      (reinterpret_cast<uintptr_t>(C) & 15))
    return 0;
  if (((long)(K - 1) * lda + M) * 4 >= 0x7FFFFFF0L || ((long)(K - 1) * ldb + N) * 4 >= 0x7FFFFFF0L || ((long)(M - 1) * ldc + N) * 4 >= 0x7FFFFFF0L) return 0;
  if (M < 64 || N < 64 || K < 256 || act == RE2E_ACT_SIGMOID_MASK_MUL) return 0;
  if ((bias && (reinterpret_cast<uintptr_t>(bias) & 15)) || (bias2 && (reinterpret_cast<uintptr_t>(bias2) & 15))) return 0;
  const NtPlan pl = nt2_plan(M, N, K, re2e_stream_is_filler(st), true);
  if (!pl.variant) return 0;
  NtArgs a;
  memset(&a, 0, sizeof(a));
  if (pl.g_sk) {
    if (!ws || wsb < pl.bytes || (long)pl.ntm * pl.ntn - pl.n_dp > POOL_SLICE_INTS) return 0;
    a.counters = ticket_slice();
    if (!a.counters) return 0;
    a.slabs = (float*)ws;
  }
  a.A = A; a.B = B; a.C = C;
  a.a_bytes = (unsigned)(((long)(K - 1) * lda + M) * 4); a.b_bytes = (unsigned)(((long)(K - 1) * ldb + N) * 4);
  a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias = bias; a.bias2 = bias2; a.act = act; a.beta = beta;
  a.ntm = pl.ntm; a.ntn = pl.ntn; a.n_dp = pl.n_dp; a.g_sk = pl.g_sk; a.nkt = pl.nkt;
  static const bool nomem = exp_env("RE2E_IGEMM_NOMEM") != nullptr;
  a.nomem = nomem ? 1 : 0;

  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;
  if (log_calls)
    fprintf(stderr, "[igemm] A=DenseM B=DenseM tile=%dx%dx%d vec=1 M=%d N=%d K=%d splits=%d\n", pl.bm, pl.bn, pl.bk, M, N, K, pl.g_sk ? -pl.g_sk : 1);
  if (exp_env("RE2E_NT2_LOG")) fprintf(stderr, "[tn2] %dx%dx%d variant %d tiles %ld dp %d sk %d est %.1f us\n", M, N, K, pl.variant, (long)pl.ntm * pl.ntn, pl.n_dp, pl.g_sk, pl.est * 1e6);
  switch (pl.variant) {
    case 6: nt2_launch<T128x128k16s3, 1>(a, pl, st); break;
    default: return 0;
  }
  return 1;
}

