// A torch-free caller of the C ABI (include/mmu_hip.h): what a native host would link against libmmu_hip.so.
// y = x W^T through mmu_gemm_bf16 (layout 0 = nn.Linear forward) on integer-valued bf16 data, checked exactly
// against a CPU loop; then C += x W^T through the accumulate epilogue.  Exit code 0 = pass.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mmu_hip.h"

static uint16_t f2bf(float f) {  // exact for the small integers used here
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
#define CK(x)                                                              \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) {                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));        \
            return 2;                                                      \
        }                                                                  \
    } while (0)

int main() {
    if (mmu_abi_version() < 3) return 3;  // the entry points used here exist since version 3; later versions only add
    const int M = 520, N = 264, K = 392;  // ragged against every tile size
    std::vector<uint16_t> A((size_t)M * K), B((size_t)N * K), C((size_t)M * N), C0((size_t)M * N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 16) % 5) - 2; };
    for (auto& v : A) v = f2bf((float)rnd());
    for (auto& v : B) v = f2bf((float)rnd());
    for (auto& v : C0) v = f2bf((float)rnd());
    void *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 2));
    CK(hipMalloc(&dB, B.size() * 2));
    CK(hipMalloc(&dC, C.size() * 2));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int rc = mmu_gemm_bf16(0, dA, dB, dC, nullptr, nullptr, M, N, K, K, K, N, 0, 0, 0, st);
    if (rc) { fprintf(stderr, "mmu_gemm_bf16 -> %d\n", rc); return 4; }
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(C.data(), dC, C.size() * 2, hipMemcpyDeviceToHost));
    std::vector<float> ref((size_t)M * N);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc += bf2f(A[(size_t)m * K + k]) * bf2f(B[(size_t)n * K + k]);
            ref[(size_t)m * N + n] = acc;
        }
    auto round_bf = [](float f) {  // round-to-nearest-even to bf16
        uint32_t u;
        std::memcpy(&u, &f, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    for (size_t i = 0; i < C.size(); ++i)
        if (C[i] != round_bf(ref[i])) { fprintf(stderr, "store epilogue mismatch at %zu\n", i); return 5; }
    CK(hipMemcpy(dC, C0.data(), C0.size() * 2, hipMemcpyHostToDevice));
    rc = mmu_gemm_bf16(0, dA, dB, dC, nullptr, nullptr, M, N, K, K, K, N, 0, 0, /*accumulate=*/1, st);
    if (rc) return 6;
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(C.data(), dC, C.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < C.size(); ++i)
        if (C[i] != round_bf(ref[i] + bf2f(C0[i]))) { fprintf(stderr, "accumulate epilogue mismatch at %zu\n", i); return 7; }
    // argument validation: unsupported arguments are refused with -1, nothing is launched
    if (mmu_gemm_bf16(0, dA, dB, dC, nullptr, nullptr, M, N + 1, K, K, K, N, 0, 0, 0, st) != -1) return 8;
    printf("c-abi caller ok: %dx%dx%d store + accumulate exact\n", M, N, K);
    return 0;
}
