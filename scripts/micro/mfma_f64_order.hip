// In which order does v_mfma_f64_16x16x4_f64 add its four products to the accumulator, and with which
// roundings?  D = A (16 x 4) B (4 x 16) + C on random operands of mixed magnitude, against host chains
//   up:   d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))     (k = 0 first)
//   down: k = 3 first
// and unfused / pairwise variants.  A bit-exact direct-form FIR on the matrix pipe needs ONE of the two
// chains, exactly.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_f64_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void mfma_kernel(const double *A, const double *B, const double *C, double *D, int chain)
{
    // A[i][k] row-major 16 x 4, B[k][j] row-major 4 x 16, C / D[lane][r] raw (layout probed on the host)
    const int l = threadIdx.x;
    v4d c;
    for (int r = 0; r < 4; ++r)
        c[r] = C[l * 4 + r];
    for (int t = 0; t < chain; ++t) {  // `chain` dependent instructions on the same accumulator
        const double a = A[t * 64 + (l % 16) * 4 + l / 16];
        const double b = B[t * 64 + (l / 16) * 16 + l % 16];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r)
        D[l * 4 + r] = c[r];
}

static uint64_t bits(double x)
{
    uint64_t u;
    std::memcpy(&u, &x, 8);
    return u;
}

int main()
{
    const int chain = 3;
    std::mt19937_64 rng(12345);
    auto rnd = [&](int spread) {
        const double m = (double)(rng() >> 11) * 0x1p-53 * 2.0 - 1.0;
        return std::ldexp(m, (int)(rng() % (2 * spread + 1)) - spread);
    };
    double *dA, *dB, *dC, *dD;
    (void)hipMalloc(&dA, 8 * 64 * chain);
    (void)hipMalloc(&dB, 8 * 64 * chain);
    (void)hipMalloc(&dC, 8 * 256);
    (void)hipMalloc(&dD, 8 * 256);
    // ---- layout probe: A = e_i e_0^T scaled, exact small integers
    std::vector<double> A(64 * chain, 0.0), B(64 * chain, 0.0), C(256, 0.0), D(256);
    for (int i = 0; i < 16; ++i)
        A[i * 4 + 0] = i + 1;
    for (int j = 0; j < 16; ++j)
        B[0 * 16 + j] = 100 * (j + 1);
    (void)hipMemcpy(dA, A.data(), 8 * A.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), 8 * B.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), 8 * 256, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(dA, dB, dC, dD, 1);
    (void)hipMemcpy(D.data(), dD, 8 * 256, hipMemcpyDeviceToHost);
    int row_of[256], col_of[256];
    bool ok = true;
    for (int s = 0; s < 256; ++s) {
        const int v = (int)D[s];
        col_of[s] = v / 100 / ((v % 100) ? 1 : 1);
        // v = (i + 1) * 100 * (j + 1): factor it
        int fi = -1, fj = -1;
        for (int i = 1; i <= 16 && fi < 0; ++i)
            for (int j = 1; j <= 16; ++j)
                if (i * 100 * j == v && (fi < 0)) {
                    // ambiguous products exist (2*3 = 3*2): disambiguate with the documented column = lane % 16
                    if (j - 1 == (s / 4) % 16) {
                        fi = i - 1;
                        fj = j - 1;
                    }
                }
        row_of[s] = fi;
        col_of[s] = fj;
        ok = ok && fi >= 0;
    }
    std::printf("layout probe %s: lane 0 rows %d %d %d %d, lane 16 rows %d %d %d %d, lane 17 col %d\n", ok ? "ok" : "FAILED", row_of[0], row_of[1],
                row_of[2], row_of[3], row_of[64], row_of[65], row_of[66], row_of[67], col_of[17 * 4]);
    if (!ok)
        return 1;
    // ---- the order test
    long up = 0, down = 0, unfused_up = 0, pair = 0, total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        const int spread = 1 + trial % 40;
        for (auto &v : A)
            v = rnd(spread);
        for (auto &v : B)
            v = rnd(spread);
        for (auto &v : C)
            v = rnd(spread);
        (void)hipMemcpy(dA, A.data(), 8 * A.size(), hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), 8 * B.size(), hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, C.data(), 8 * 256, hipMemcpyHostToDevice);
        mfma_kernel<<<1, 64>>>(dA, dB, dC, dD, chain);
        (void)hipMemcpy(D.data(), dD, 8 * 256, hipMemcpyDeviceToHost);
        for (int s = 0; s < 256; ++s) {
            const int i = row_of[s], j = col_of[s];
            double u = C[s], d = C[s], uu = C[s], p = C[s];
            for (int t = 0; t < chain; ++t) {
                const double *a = &A[t * 64 + i * 4];
                const double *b = &B[t * 64];
                for (int k = 0; k < 4; ++k)
                    u = std::fma(a[k], b[k * 16 + j], u);
                for (int k = 3; k >= 0; --k)
                    d = std::fma(a[k], b[k * 16 + j], d);
                for (int k = 0; k < 4; ++k)
                    uu = uu + a[k] * b[k * 16 + j];
                p = p + ((a[0] * b[j] + a[1] * b[16 + j]) + (a[2] * b[32 + j] + a[3] * b[48 + j]));
            }
            up += bits(u) == bits(D[s]);
            down += bits(d) == bits(D[s]);
            unfused_up += bits(uu) == bits(D[s]);
            pair += bits(p) == bits(D[s]);
            ++total;
        }
    }
    std::printf("of %ld results (chains of %d instructions): fma chain k up %ld, k down %ld, unfused up %ld, pairwise %ld\n", total, chain, up, down,
                unfused_up, pair);
    return 0;
}
