// Probe of ds_read_b64_tr_b16 (gfx950): which element lands where.  Image: 64 rows x 40 columns of 16-bit values, value =
// row * 100 + col.  Group g (16 lanes) reads the block rows 4g..4g+3, cols 0..15: lane 4q+p supplies (row 4g+q, col 4p).
// Expected (guide T10): lane i of the group receives column i, element q = row 4g+q.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/tr_read_probe.hip -o tools/micro/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(const short* in, short* out) {
    __shared__ __attribute__((aligned(16))) short img[64 * 40];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 40; i += 64) img[i] = in[i];
    __syncthreads();
    const int li = l & 15, q = li >> 2, p = li & 3, g = l >> 4;
    const short* a = &img[(4 * g + q) * 40 + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    std::vector<short> h(64 * 40), o(256);
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c < 40; ++c) h[r * 40 + c] = (short)(r * 100 + c);
    short *d, *e;
    hipMalloc(&d, h.size() * 2);
    hipMalloc(&e, 512);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o.data(), e, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        for (int j = 0; j < 4; ++j)
            if (o[l * 4 + j] != (short)((4 * g + j) * 100 + i)) ++bad;
    }
    for (int l = 0; l < 20; ++l) printf("lane %2d: %d %d %d %d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    printf("mismatches vs guide T10 expectation: %d\n", bad);
    return bad != 0;
}
