// Does gfx950 honour the VOP3P `clamp` output modifier on packed-f32 instructions (result clamped to [0, 1], NaN -> 0 with
// DX10_CLAMP)?   hipcc --offload-arch=gfx950 -O2 tools/clamp_probe.hip -o tools/clamp_probe && ./tools/clamp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const f32x2* a, const f32x2* b, const f32x2* c, f32x2* o_fma, f32x2* o_mul, f32x2* o_add) {
  const int i = threadIdx.x;
  f32x2 r, r2, r3;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
  asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r2) : "v"(a[i]), "v"(b[i]));
  asm volatile("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r3) : "v"(a[i]), "v"(c[i]));
  o_fma[i] = r; o_mul[i] = r2; o_add[i] = r3;
}
int main() {
  const int n = 64;
  f32x2 ha[n], hb[n], hc[n], *da, *db, *dc, *o1, *o2, *o3, h1[n], h2[n], h3[n];
  const float vals[] = {-3.f, -1.f, -1e-30f, -0.f, 0.f, 1e-30f, 0.25f, 0.5f, 1.f, 1.5f, 7.f, 1e30f, -1e30f, NAN, INFINITY, -INFINITY};
  for (int i = 0; i < n; ++i) {
    ha[i] = f32x2{vals[i % 16], vals[(i * 7 + 3) % 16]};
    hb[i] = f32x2{(i & 1) ? 0.5f : 2.f, (i & 2) ? 1.f : -1.f};
    hc[i] = f32x2{vals[(i * 5 + 1) % 16] * 0.5f, 0.125f};
  }
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof ha); hipMalloc(&dc, sizeof ha);
  hipMalloc(&o1, sizeof ha); hipMalloc(&o2, sizeof ha); hipMalloc(&o3, sizeof ha);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof ha, hipMemcpyHostToDevice);
  hipMemcpy(dc, hc, sizeof ha, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(n), 0, 0, da, db, dc, o1, o2, o3);
  hipMemcpy(h1, o1, sizeof ha, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, sizeof ha, hipMemcpyDeviceToHost);
  hipMemcpy(h3, o3, sizeof ha, hipMemcpyDeviceToHost);
  auto cl = [](float v) { return std::isnan(v) ? 0.f : v < 0.f ? 0.f : v > 1.f ? 1.f : v; };
  int bad = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 2; ++j) {
      const float e1 = cl(fmaf(ha[i][j], hb[i][j], hc[i][j])), e2 = cl(ha[i][j] * hb[i][j]), e3 = cl(ha[i][j] + hc[i][j]);
      if (h1[i][j] != e1 || h2[i][j] != e2 || h3[i][j] != e3) {
        ++bad;
        printf("lane %d.%d a=%g b=%g c=%g  fma %g (want %g)  mul %g (want %g)  add %g (want %g)\n", i, j, ha[i][j], hb[i][j], hc[i][j],
               h1[i][j], e1, h2[i][j], e2, h3[i][j], e3);
      }
    }
  printf("clamp probe: %d mismatches of %d\n", bad, 2 * n);
  return bad != 0;
}
