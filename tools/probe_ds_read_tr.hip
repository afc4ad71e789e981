// probe of ds_read_b64_tr_b16 semantics: lds[i] = i, lane l supplies the address of el16 element 4*l (8-byte pieces in lane order)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    __attribute__((address_space(3))) v4s* p = (__attribute__((address_space(3))) v4s*)(lds + 4 * l);
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    short h[4096], o[256];
    for (int i = 0; i < 4096; ++i) h[i] = (short)i;
    short *din, *dout;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, sizeof(o));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    return 0;
}
