// Times hgv_gemm_nt16 / hgv_gemm_tn16 on one shape from a stand-alone binary; the kernel source is compiled in
// (-DKSRC=<file>), so that sed-made variants of gemm_bf16.hip (ablations) can be timed side by side.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Ihgvae_amd/csrc -Iinclude -DKSRC='"..."' tools/lab/nt16_lab.hip
#include KSRC
#include <vector>
#include <stdlib.h>
static char g_err[512];
void hgv_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 160984;
    const int K1 = argc > 2 ? atoi(argv[2]) : 138, K2 = argc > 3 ? atoi(argv[3]) : 138, N = argc > 4 ? atoi(argv[4]) : 138;
    const char* tag = argc > 5 ? argv[5] : "base";
    const int64_t lda = (K1 + 3) / 4 * 4, ldb = (K2 + 3) / 4 * 4, ldc = (N + 3) / 4 * 4;
    std::vector<float> h((size_t)M * (lda > ldc ? (lda > ldb ? lda : ldb) : (ldc > ldb ? ldc : ldb)) + 1024);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    // SPARSE=p: a fraction p of G is zero, as in a gradient behind relu and dropout (drawn before the first HIP call: the
    // runtime's own threads use rand() as well)
    const double sp = getenv("SPARSE") ? atof(getenv("SPARSE")) : 0.0;
    std::vector<float> hg(h.begin(), h.begin() + (size_t)M * ldc);
    if (sp > 0) for (auto& v : hg) if ((double)rand() / RAND_MAX < sp) v = 0.f;
    float *A1, *A2 = nullptr, *W, *C, *bias, *G, *GW1, *GW2;
    hipMalloc(&A1, (size_t)M * lda * 4); hipMemcpy(A1, h.data(), (size_t)M * lda * 4, hipMemcpyHostToDevice);
    if (K2) { hipMalloc(&A2, (size_t)M * ldb * 4); hipMemcpy(A2, h.data(), (size_t)M * ldb * 4, hipMemcpyHostToDevice); }
    hipMalloc(&G, (size_t)M * ldc * 4); hipMemcpy(G, hg.data(), (size_t)M * ldc * 4, hipMemcpyHostToDevice);
    hipMalloc(&W, (size_t)N * (K1 + K2) * 4); hipMemcpy(W, h.data(), (size_t)N * (K1 + K2) * 4, hipMemcpyHostToDevice);
    hipMalloc(&bias, N * 4); hipMemcpy(bias, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMalloc(&C, (size_t)M * ldc * 4);
    hipMalloc(&GW1, (size_t)N * (K1 + K2) * 4); GW2 = GW1 + K1;
    void* img; hipMalloc(&img, hgv_pack16_bytes(N, K1, K2));
    if (hgv_pack16_weights(W, K1 + K2, 1, K1, K2 ? W + K1 : nullptr, K1 + K2, 1, K2, N, img, 0)) { printf("pack: %s\n", g_err); return 1; }
    void* ws; size_t wsb = hgv_gemm_tn16_ws(M, N, K1, K2); hipMalloc(&ws, wsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const bool do_tn = getenv("TN") != nullptr;
    const int rep = getenv("REP") ? atoi(getenv("REP")) : 1;      // launches per timed interval (queued back to back)
    float best = 1e9, sum = 0; const int iters = 20;
    for (int it = 0; it < iters + 3; ++it) {
        hipEventRecord(e0, 0);
        int rc = 0;
        for (int r = 0; r < rep && !rc; ++r) rc = do_tn ? hgv_gemm_tn16(G, ldc, N, A1, lda, K1, GW1, K1 + K2, A2, ldb, K2, GW2, K1 + K2, M, bias, 0, ws, wsb, 0)
                       : hgv_gemm_nt16(A1, lda, K1, A2, ldb, K2, img, bias, M, N, C, ldc, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        if (rc) { printf("call: %s\n", g_err); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    unsigned long long hash = 1469598103934665603ull;          // FNV-1a over the output: variants must agree bit for bit
    if (!do_tn) {
        std::vector<unsigned> out((size_t)M * ldc);
        hipMemcpy(out.data(), C, out.size() * 4, hipMemcpyDeviceToHost);
        for (int64_t r = 0; r < M; ++r)
            for (int c = 0; c < N; ++c) { hash ^= out[(size_t)r * ldc + c]; hash *= 1099511628211ull; }
    }
    if (do_tn) {
        std::vector<unsigned> out((size_t)N * (K1 + K2));
        hipMemcpy(out.data(), GW1, out.size() * 4, hipMemcpyDeviceToHost);
        for (unsigned v : out) { hash ^= v; hash *= 1099511628211ull; }
    }
    const double fl = 2.0 * M * (K1 + K2) * N;
    printf("%-14s %s M=%lld K=%d+%d N=%d: avg %.1f us  best %.1f us  %.1f TF/s  out %016llx\n", tag, do_tn ? "TN" : "NT", (long long)M, K1, K2, N,
           sum / iters * 1e3, best * 1e3, fl / (sum / iters * 1e-3) / 1e12, hash);
    return 0;
}
