// Raw buffer loads that straddle the end of the resource: is the range check per dword (the in-range dword
// arrives, the other reads 0) or per access (everything 0)?  Decides whether a channel "pair" whose second
// member lies past the end of a Line may be loaded as one 8-byte piece (odd channel counts in the overlap-save
// FIR).  Same resource word 3 as the library (0x00020000).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ void probe(const float *base, int nrec_bytes, float *out, int step)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, nrec_bytes, 0x00020000);
    const int l = threadIdx.x;
    // lane l loads 8 bytes at byte offset 4 l: the last in-range lane straddles the end
    const v2u a = __builtin_amdgcn_raw_buffer_load_b64(rs, step * l, 0, 0);
    const float2 af = __builtin_bit_cast(float2, a);
    out[2 * l] = af.x;
    out[2 * l + 1] = af.y;
    const v4u b = __builtin_amdgcn_raw_buffer_load_b128(rs, step * l, 0, 0);
    const float4 bf = __builtin_bit_cast(float4, b);
    out[128 + 4 * l] = bf.x;
    out[128 + 4 * l + 1] = bf.y;
    out[128 + 4 * l + 2] = bf.z;
    out[128 + 4 * l + 3] = bf.w;
}

int main()
{
    float h[64], *d, *o, r[128 + 256];
    for (int i = 0; i < 64; ++i)
        h[i] = 100.0f + i;
    (void)hipMalloc(&d, sizeof h);
    (void)hipMalloc(&o, sizeof r);
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int step : {4, 8}) {
        const int n = 11;  // the resource covers 11 floats (44 bytes) of the 64 that exist
        probe<<<1, 64>>>(d, 4 * n, o, step);
        (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        std::printf("b64 at offsets %d l, num_records = %d bytes:\n", step, 4 * n);
        for (int l = 3; l < 12; ++l)
            std::printf("  lane %2d (offset %2d): %6.1f %6.1f\n", l, step * l, r[2 * l], r[2 * l + 1]);
        std::printf("b128:\n");
        for (int l = 1; l < 12; ++l)
            std::printf("  lane %2d (offset %2d): %6.1f %6.1f %6.1f %6.1f\n", l, step * l, r[128 + 4 * l], r[128 + 4 * l + 1], r[128 + 4 * l + 2], r[128 + 4 * l + 3]);
    }
    return 0;
}
