// Issue rate of packed-f32 instructions on gfx950 with and without op_sel broadcast / clamp: cycles per wave-instruction with one
// wave per SIMD.   hipcc --offload-arch=gfx950 -O2 tools/pk_rate_probe.hip -o tools/pk_rate_probe && ./tools/pk_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHAIN8(OP)                                                                                   \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                     \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
               : "v"(x), "v"(y))
#define FMA_PLAIN(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define FMA_SELHI(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " op_sel_hi:[0,1,1]\n\t"
#define FMA_SEL(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " op_sel:[1,0,0]\n\t"
#define FMA_CLAMP(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " clamp\n\t"
#define ADD_PLAIN(i) "v_pk_add_f32 %" #i ", %8, %" #i "\n\t"
#define ADD_SELHI(i) "v_pk_add_f32 %" #i ", %8, %" #i " op_sel_hi:[0,1]\n\t"
#define ADD_SEL(i) "v_pk_add_f32 %" #i ", %8, %" #i " op_sel:[1,0]\n\t"
#define ADD_CLAMP(i) "v_pk_add_f32 %" #i ", %8, %" #i " clamp\n\t"
template <int V>
__global__ void probe(long long* out, float seed) {
  f32x2 a[8], x = {seed, seed * 0.5f}, y = {0.999f, 1.001f};
  for (int i = 0; i < 8; ++i) a[i] = f32x2{seed + i, seed - i};
  const long long t0 = clock64();
  for (int it = 0; it < 512; ++it) {
    if (V == 0) CHAIN8(FMA_PLAIN);
    if (V == 1) CHAIN8(FMA_SELHI);
    if (V == 2) CHAIN8(FMA_SEL);
    if (V == 3) CHAIN8(FMA_CLAMP);
    if (V == 4) CHAIN8(ADD_PLAIN);
    if (V == 5) CHAIN8(ADD_SELHI);
    if (V == 6) CHAIN8(ADD_SEL);
    if (V == 7) CHAIN8(ADD_CLAMP);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}
template <int V>
void run(const char* name, long long* d, int waves_per_simd) {
  long long h[2];
  hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, 1.0f);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%-34s %d wave(s)/SIMD: %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, (double)h[0] / (512.0 * 8.0 * waves_per_simd));
}
int main() {
  long long* d;
  (void)hipMalloc(&d, 16);
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_pk_fma_f32", d, w);
    run<1>("v_pk_fma_f32 op_sel_hi:[0,1,1]", d, w);
    run<2>("v_pk_fma_f32 op_sel:[1,0,0]", d, w);
    run<3>("v_pk_fma_f32 clamp", d, w);
    run<4>("v_pk_add_f32", d, w);
    run<5>("v_pk_add_f32 op_sel_hi:[0,1]", d, w);
    run<6>("v_pk_add_f32 op_sel:[1,0]", d, w);
    run<7>("v_pk_add_f32 clamp", d, w);
  }
  return 0;
}
