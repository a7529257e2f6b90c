// Stand-alone laboratory for the selective-scan kernels: includes csrc/selective_scan.hip as is, runs it at the BASELINE
// configs[1] size without torch, checks a few (row, channel) columns against a double-precision host recurrence and times
// the launches with HIP events.  Lets kernel variants (-D switches) be compared in seconds on a gpurun box.
// Build (here or on the box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/micro/sscan_lab.hip -o tools/micro/bin/sscan_lab
// Run: sscan_lab [B=64] [L=1043] [Di=512] [N=32] [reps=20] [bwd=1] [time segments: 0 auto, 1 off, k] [forward edition 2 | 3]
#include "../../recurrent-offpolicy-rl_amd/csrc/selective_scan.hip"
#include "../../recurrent-offpolicy-rl_amd/csrc/misc.hip"          // the per-dispatch timing registry the launches refer to
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev(const std::vector<float>& h) {
    float* d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return d;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 1043, Di = argc > 3 ? atoi(argv[3]) : 512,
              N = argc > 4 ? atoi(argv[4]) : 32, reps = argc > 5 ? atoi(argv[5]) : 20, do_bwd = argc > 6 ? atoi(argv[6]) : 1;
    const int R = 16, ldx = 2 * Di, ldb = R + 2 * N;
    const int tseg = argc > 7 ? atoi(argv[7]) : 0;
    if (argc > 8) { resel_selective_scan_fwd_edition(atoi(argv[8])); printf("forward edition %d\n", atoi(argv[8])); }
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> xz((size_t)B * L * ldx), xdbl((size_t)B * L * ldb), delta((size_t)B * L * Di), A((size_t)Di * N), Dp(Di), db(Di),
        start((size_t)B * L, 0.f), dout((size_t)B * L * Di);
    for (auto& v : xz) v = nd(g);
    for (auto& v : xdbl) v = nd(g);
    for (auto& v : delta) v = 0.5f * nd(g);
    for (auto& v : A) v = -std::exp(0.3f * nd(g));
    for (auto& v : Dp) v = nd(g);
    for (auto& v : db) v = 0.1f * nd(g);
    for (auto& v : dout) v = nd(g);
    for (int b = 0; b < B; ++b) {
        for (int t = 0; t < 18 && t < L; ++t) start[(size_t)b * L + t] = 1.f;
        if (L > 400) start[(size_t)b * L + 300 + b] = 1.f;           // one mid-row reset per row
    }
    float *d_xz = dev(xz), *d_xdbl = dev(xdbl), *d_delta = dev(delta), *d_A = dev(A), *d_D = dev(Dp), *d_db = dev(db), *d_start = dev(start),
          *d_dout = dev(dout);
    float *d_out, *d_ckpt, *d_dxz, *d_ddelta, *d_dxdbl, *d_dA, *d_dD, *d_ddb;
    void* d_ws;
    CK(hipMalloc(&d_out, (size_t)B * L * Di * 4));
    const size_t ckb = resel_selective_scan_ckpt_bytes(B, L, Di, N);
    CK(hipMalloc(&d_ckpt, ckb ? ckb : 16));
    CK(hipMalloc(&d_dxz, (size_t)B * L * ldx * 4));
    CK(hipMalloc(&d_ddelta, (size_t)B * L * Di * 4));
    CK(hipMalloc(&d_dxdbl, (size_t)B * L * ldb * 4));
    CK(hipMalloc(&d_dA, (size_t)Di * N * 4)); CK(hipMalloc(&d_dD, Di * 4)); CK(hipMalloc(&d_ddb, Di * 4));
    CK(hipMalloc(&d_ws, resel_selective_scan_bwd_workspace_bytes(B, L, Di, N, argc > 7 ? atoi(argv[7]) : 0)));
    hipStream_t s = 0;
    void* d_fws = nullptr;
    { const size_t nb = resel_selective_scan_fwd_workspace_bytes(B, L, Di, N, tseg); if (nb) CK(hipMalloc(&d_fws, nb)); printf("forward time segments workspace: %zu bytes\n", nb); }
    auto fwd = [&]() {
        return resel_selective_scan_fwd(d_xz, ldx, d_delta, Di, d_xz + Di, ldx, d_A, d_xdbl + R, ldb, d_xdbl + R + N, ldb, d_D, d_db, d_start,
                                        d_out, Di, d_ckpt, nullptr, d_fws, B, L, Di, N, 1, tseg, nullptr, 0u, s);
    };
    auto bwd = [&]() {
        return resel_selective_scan_bwd(d_xz, ldx, d_delta, Di, d_xz + Di, ldx, d_A, d_xdbl + R, ldb, d_xdbl + R + N, ldb, d_D, d_db, d_start,
                                        d_dout, Di, d_ckpt, d_dxz, ldx, d_ddelta, Di, d_dxz + Di, ldx, d_dxdbl + R, ldb, d_dxdbl + R + N, ldb,
                                        d_dA, d_dD, d_ddb, d_ws, B, L, Di, N, 1, tseg, nullptr, nullptr, 0u, s);
    };
#ifdef SSCAN_STAMP
    const size_t nst = (size_t)((B + 7) / 8 * 8) * ((Di + 63) / 64) * 8 * 8;
    CK(hipMalloc(&g_stamps, nst * 8));
    CK(hipMemset(g_stamps, 0, nst * 8));
#endif
    int rc = fwd();
    CK(hipDeviceSynchronize());
#ifdef SSCAN_STAMP
    {
        fwd(); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs(nst);
        CK(hipMemcpy(hs.data(), g_stamps, nst * 8, hipMemcpyDeviceToHost));
        double sum[5] = {0, 0, 0, 0, 0}, ct = 0, cr = 0; size_t nw = 0;
        for (size_t i = 0; i < nst; i += 8) if (hs[i + 2]) { for (int k = 0; k < 5; ++k) sum[k] += (double)hs[i + k]; ct += hs[i + 5]; cr += hs[i + 6]; ++nw; }
        {
            unsigned long long r_min = ~0ull, e_max = 0; std::vector<double> starts;
            for (size_t i = 0; i < nst; i += 8) if (hs[i + 2]) { r_min = std::min(r_min, hs[i + 7]); e_max = std::max(e_max, hs[i + 7] + hs[i + 6]); }
            int late = 0; double latest = 0;
            for (size_t i = 0; i < nst; i += 8) if (hs[i + 2]) { const double st = (hs[i + 7] - r_min) * 0.01; latest = std::max(latest, st); if (st > 20) ++late; }
            printf("first wave start -> last wave end %.1f us; waves starting > 20 us after the first: %d; latest start +%.1f us\n", (e_max - r_min) * 0.01, late, latest);
        }
        {
            double xs[8] = {0}, xc[8] = {0}, xmax[8] = {0}, xclk_t[8] = {0}, xclk_r[8] = {0};
            const size_t per_blk = 8 * 8;                 // 8 wave slots x 8 words (4 used waves)
            for (size_t i = 0; i < nst; i += 8) if (hs[i + 2]) {
                const int x = (int)((i / per_blk) & 7);
                xs[x] += hs[i + 6] * 0.01; xc[x] += 1; xmax[x] = std::max(xmax[x], hs[i + 6] * 0.01); xclk_t[x] += hs[i + 5]; xclk_r[x] += hs[i + 6];
            }
            for (int x = 0; x < 8; ++x) printf("  blockIdx%%8=%d: mean body %.1f us, max %.1f us, clock %.2f GHz\n", x, xs[x] / xc[x], xmax[x], xclk_t[x] / xclk_r[x] * 0.1);
        }
        {
            int hist[64] = {0};
            for (size_t i = 0; i < nst; i += 8) if (hs[i + 2]) { int b = (int)(hs[i + 6] * 0.01 / 10); if (b > 63) b = 63; ++hist[b]; }
            printf("  wave body duration histogram (10 us bins): ");
            for (int b = 0; b < 64; ++b) if (hist[b]) printf("[%d-%d us: %d] ", b * 10, b * 10 + 10, hist[b]);
            printf("\n");
        }
        printf("in-kernel clock %.2f GHz (memtime / memrealtime); kernel body %.1f us\n", ct / cr * 0.1, cr / nw * 0.01);
        const char* nm[5] = {"stage", "barrier1", "scan", "barrier2", "output"};
        double tot = 0; for (int k = 0; k < 5; ++k) tot += sum[k];
        printf("stamps over %zu waves: ", nw);
        for (int k = 0; k < 5; ++k) printf("%s %.0f cyc (%.1f %%)  ", nm[k], sum[k] / nw, 100 * sum[k] / tot);
        printf("| total %.0f cycles per wave\n", tot / nw);
    }
#endif
    if (rc) { printf("fwd rc %d\n", rc); return 1; }
    // ---- host check of a few columns (double precision)
    std::vector<float> out((size_t)B * L * Di);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    double max_err = 0, max_ref = 0;
    const int bs[3] = {0, B / 2, B - 1}, ds[4] = {0, 1, Di / 2 + 3, Di - 1};
    for (int b : bs) for (int d : ds) {
        std::vector<double> h(N, 0.0);
        for (int t = 0; t < L; ++t) {
            const size_t tok = (size_t)b * L + t;
            double dl = delta[tok * Di + d] + db[d];
            dl = dl > 20 ? dl : std::log1p(std::exp(dl));
            const double u = xz[tok * ldx + d], z = xz[tok * ldx + Di + d];
            double y = 0;
            for (int n = 0; n < N; ++n) {
                const double dA = start[tok] != 0.f ? 0.0 : std::exp(dl * A[(size_t)d * N + n]);
                h[n] = dA * h[n] + dl * u * xdbl[tok * ldb + R + n];
                y += h[n] * xdbl[tok * ldb + R + N + n];
            }
            y = (y + Dp[d] * u) * (z / (1 + std::exp(-z)));
            max_err = std::max(max_err, std::fabs(y - (double)out[tok * Di + d]));
            max_ref = std::max(max_ref, std::fabs(y));
        }
    }
    printf("fwd check: max |err| %.3e  (max |ref| %.3e) -> %s\n", max_err, max_ref, max_err <= 1e-4 * max_ref ? "OK" : "MISMATCH");
    double cks = 0;
    for (size_t i = 0; i < out.size(); i += 97) cks += out[i];
    printf("fwd checksum %.6f\n", cks);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) fwd();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) fwd();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fb = 4.0 * B * Di * L * 4 + 4.0 * B * N * L * 2 + (double)B * L;
    printf("fwd: %.1f us  -> %.2f TB/s algorithmic (%.1f %% of 8 TB/s)\n", ms * 1e3 / reps, fb / (ms * 1e-3 / reps) / 1e12, fb / (ms * 1e-3 / reps) / 8e12 * 100);
    if (do_bwd) {
        rc = bwd();
        CK(hipDeviceSynchronize());
        if (rc) { printf("bwd rc %d\n", rc); return 1; }
        std::vector<float> dxz((size_t)B * L * ldx), dde((size_t)B * L * Di), dxd((size_t)B * L * ldb), dA((size_t)Di * N);
        CK(hipMemcpy(dxz.data(), d_dxz, dxz.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dde.data(), d_ddelta, dde.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dxd.data(), d_dxdbl, dxd.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dA.data(), d_dA, dA.size() * 4, hipMemcpyDeviceToHost));
        double c1 = 0, c2 = 0, c3 = 0, c4 = 0;
        for (size_t i = 0; i < dxz.size(); i += 101) c1 += dxz[i];
        for (size_t i = 0; i < dde.size(); i += 101) c2 += dde[i];
        for (size_t tok = 0; tok < (size_t)B * L; tok += 7) for (int n = 0; n < 2 * N; ++n) c3 += dxd[tok * ldb + R + n];
        for (size_t i = 0; i < dA.size(); ++i) c4 += dA[i];
        std::vector<float> dDv(Di), ddbv(Di);
        CK(hipMemcpy(dDv.data(), d_dD, Di * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ddbv.data(), d_ddb, Di * 4, hipMemcpyDeviceToHost));
        double c5 = 0, c6 = 0; for (int i = 0; i < Di; ++i) { c5 += dDv[i]; c6 += ddbv[i]; }
        printf("bwd checksums: dxz %.6f ddelta %.6f dBC %.6f dA %.6f dD %.6f dbias %.6f\n", c1, c2, c3, c4, c5, c6);
        for (int i = 0; i < 2; ++i) bwd();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) bwd();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bb = 4.0 * B * Di * L * 7 + 4.0 * B * N * L * 4;
        printf("bwd (all launches): %.1f us  -> %.2f TB/s algorithmic (%.1f %% of 8 TB/s)\n", ms * 1e3 / reps, bb / (ms * 1e-3 / reps) / 1e12,
               bb / (ms * 1e-3 / reps) / 8e12 * 100);
    }
    return 0;
}
