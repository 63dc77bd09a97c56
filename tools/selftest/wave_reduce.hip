// Stand-alone check of tfc_wave_reduce16 / tfc_wave_reduce4 (csrc/tfc_common.h) on the GPU.
// build: hipcc -O3 --offload-arch=gfx950 tools/selftest/wave_reduce.hip -o tools/selftest/wave_reduce.bin
#include "../../matten_amd/csrc/tfc_common.h"
#include <cstdio>
#include <vector>
int tfc_set_error(int code, const char*, ...) { return code; }
__global__ void k16(const float* in, float* out, int* idx) {
    const int l = threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[i * 64 + l];
    out[l] = tfc_wave_reduce16(v);
    idx[l] = tfc_reduce16_index(l);
}
__global__ void k4(const float* in, float* out, int* idx) {
    const int l = threadIdx.x;
    float v[4];
    for (int i = 0; i < 4; ++i) v[i] = in[i * 64 + l];
    out[l] = tfc_wave_reduce4(v);
    idx[l] = tfc_reduce4_index(l);
}
int main() {
    std::vector<float> h(16 * 64);
    for (int i = 0; i < 16 * 64; ++i) h[i] = (float)((i * 37) % 101) - 50.f;  // exact in fp32
    float *din, *dout;
    int* didx;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 64 * 4);
    hipMalloc(&didx, 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int which = 0; which < 2; ++which) {
        const int V = which ? 4 : 16;
        if (which) k4<<<1, 64>>>(din, dout, didx); else k16<<<1, 64>>>(din, dout, didx);
        float o[64];
        int ix[64];
        hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
        hipMemcpy(ix, didx, 256, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            float ref = 0;
            for (int j = 0; j < 64; ++j) ref += h[ix[l] * 64 + j];
            if (ix[l] < 0 || ix[l] >= V || o[l] != ref) { ++bad; printf("V=%d lane %d idx %d got %g want %g\n", V, l, ix[l], o[l], ref); }
        }
    }
    printf(bad ? "FAILED %d\n" : "wave_reduce ok\n", bad);
    return bad != 0;
}
