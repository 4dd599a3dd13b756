// Hardware-semantics probe for gfx950: checks the lane maps the kernels in this directory rely on
// (MFMA operand / accumulator layouts, accumulator-as-operand k order, ds_read_tr16_b64, global_load_lds).
// Build:  hipcc --offload-arch=gfx950 -O2 probe_gfx950.hip -o probe_gfx950 ; run on an MI355X.  Prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static inline float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// --- 1. 16x16x32: A[16][32], B[32][16] (B given as Bt[16][32]: col-major => both K-contiguous) ---------------
__global__ void k_mfma16(const unsigned short* A, const unsigned short* Bt, float* C) {
  int l = threadIdx.x;
  bf8 a = *(const bf8*)(A + (l & 15) * 32 + 8 * (l >> 4));
  bf8 b = *(const bf8*)(Bt + (l & 15) * 32 + 8 * (l >> 4));
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[((l >> 4) * 4 + j) * 16 + (l & 15)] = c[j];
}
// --- 2. 32x32x16: A[32][16], Bt[32][16] ----------------------------------------------------------------------
__global__ void k_mfma32(const unsigned short* A, const unsigned short* Bt, float* C) {
  int l = threadIdx.x;
  bf8 a = *(const bf8*)(A + (l & 31) * 16 + 8 * (l >> 5));
  bf8 b = *(const bf8*)(Bt + (l & 31) * 16 + 8 * (l >> 5));
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// --- 3. accumulator as next operand: X = A1*B1 (32x32, K=16); Y = A2 * X where A2[32][32] --------------------
//        uses X regs 8s..8s+7 as B fragment of k-step s; A2's element j of lane half h must be k = 16s+8(j>>2)+4h+(j&3)
__global__ void k_acc_operand(const unsigned short* A1, const unsigned short* B1t, const unsigned short* A2, float* Y) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf8 a = *(const bf8*)(A1 + r * 16 + 8 * h);
  bf8 b = *(const bf8*)(B1t + r * 16 + 8 * h);
  f32x16 x = {};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 y = {};
  for (int s = 0; s < 2; ++s) {
    bf8 xb, a2;
    for (int j = 0; j < 8; ++j) {
      xb[j] = (__bf16)x[8 * s + j];
      int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      a2[j] = *(const __bf16*)(A2 + r * 32 + k);
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
  }
  for (int q = 0; q < 16; ++q) Y[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = y[q];
}
// --- 4. ds_read_tr16_b64: T[k][n] tile in LDS (row stride ld elements); hypothesis:
//        lane i of 16-lane group g, giving address &T[k0 + (i>>2)][n0 + 4*(i&3)], receives T[k0+q][n0+i], q=0..3 ---
__global__ void k_tr(const unsigned short* T, unsigned short* out, int ld) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 72];
  for (int i = threadIdx.x; i < 64 * ld; i += 64) lds[i] = T[i];
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, i = l & 15;
  // group g handles rows k0 = 4g, cols n0 = 0
  const unsigned short* p = &lds[(4 * g + (i >> 2)) * ld + 4 * (i & 3)];
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  *(s16x4*)(out + l * 4) = v;
}
// --- 5. global_load_lds 16B: lane i's 16 bytes land at lds_base + 16*i; source address is per lane -------------
__global__ void k_glds(const unsigned short* G, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
  int l = threadIdx.x;           // 128 threads = 2 waves
  int w = l >> 6, lane = l & 63;
  // source permuted: lane reads chunk (lane ^ 5) of its wave's 1 KiB
  const unsigned short* src = G + w * 512 + ((lane ^ 5) * 8);
  __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + w * 512), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc.
  __syncthreads();
  for (int i = l; i < 1024; i += 128) out[i] = lds[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  srand(1);
  auto rnd = [](int n) { std::vector<unsigned short> v(n); for (auto& x : v) x = f2bf((float)((rand() % 9) - 4)); return v; };
  unsigned short *dA, *dB, *dA2, *dT, *dO16; float* dC;
  CK(hipMalloc(&dA, 65536)); CK(hipMalloc(&dB, 65536)); CK(hipMalloc(&dA2, 65536)); CK(hipMalloc(&dT, 65536)); CK(hipMalloc(&dO16, 65536)); CK(hipMalloc(&dC, 65536));
  int fails = 0;
  { // 1
    auto A = rnd(16 * 32), Bt = rnd(16 * 32); std::vector<float> C(256), R(256);
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += bf2f(A[i * 32 + k]) * bf2f(Bt[j * 32 + k]); R[i * 16 + j] = s; }
    int bad = 0; for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    printf("%s mfma_16x16x32 layout (bad=%d)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  { // 2
    auto A = rnd(32 * 16), Bt = rnd(32 * 16); std::vector<float> C(1024), R(1024);
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += bf2f(A[i * 16 + k]) * bf2f(Bt[j * 16 + k]); R[i * 32 + j] = s; }
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != R[i];
    printf("%s mfma_32x32x16 layout (bad=%d)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  { // 3
    auto rs = [](int n) { std::vector<unsigned short> v(n); for (auto& x : v) x = f2bf((float)((rand() % 5) - 2)); return v; };
    auto A1 = rs(32 * 16), B1t = rs(32 * 16), A2 = rs(32 * 32); std::vector<float> Y(1024), X(1024), R(1024);
    CK(hipMemcpy(dA, A1.data(), A1.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B1t.data(), B1t.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA2, A2.data(), A2.size() * 2, hipMemcpyHostToDevice));
    k_acc_operand<<<1, 64>>>(dA, dB, dA2, dC); CK(hipMemcpy(Y.data(), dC, 4096, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += bf2f(A1[i * 16 + k]) * bf2f(B1t[j * 16 + k]); X[i * 32 + j] = s; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += bf2f(A2[i * 32 + k]) * X[k * 32 + j]; R[i * 32 + j] = s; }
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += Y[i] != R[i];
    printf("%s accumulator-as-B-operand k order (bad=%d)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  for (int ld : {16, 64, 72}) { // 4
    std::vector<unsigned short> T(64 * ld), O(256);
    for (int i = 0; i < 64 * ld; ++i) T[i] = (unsigned short)i;   // raw bit patterns as ids
    CK(hipMemcpy(dT, T.data(), T.size() * 2, hipMemcpyHostToDevice));
    k_tr<<<1, 64>>>(dT, dO16, ld); CK(hipMemcpy(O.data(), dO16, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) { int g = l >> 4, i = l & 15; bad += O[l * 4 + q] != (unsigned short)((4 * g + q) * ld + i); }
    printf("%s ds_read_tr16_b64 hypothesis ld=%d (bad=%d)\n", bad ? "FAIL" : "PASS", ld, bad); fails += bad != 0;
    if (bad) { for (int l = 0; l < 20; ++l) printf("  lane %d: %d %d %d %d\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]); }
  }
  { // 5
    std::vector<unsigned short> G(1024), O(1024);
    for (int i = 0; i < 1024; ++i) G[i] = (unsigned short)i;
    CK(hipMemcpy(dT, G.data(), 2048, hipMemcpyHostToDevice));
    k_glds<<<1, 128>>>(dT, dO16); CK(hipMemcpy(O.data(), dO16, 2048, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int w = 0; w < 2; ++w) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e)
      bad += O[w * 512 + lane * 8 + e] != (unsigned short)(w * 512 + (lane ^ 5) * 8 + e);
    printf("%s global_load_lds lane-linear dest / per-lane source (bad=%d)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  CK(hipDeviceSynchronize());
  printf("probe done, fails=%d\n", fails);
  return fails ? 2 : 0;
}
