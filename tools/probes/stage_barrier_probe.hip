// Probe for VERDICT r05 item 6 (batch-1 decoding): what does ONE stage boundary cost INSIDE a launch when it is built without
// release / acquire fences?  A decoder stage's output is a row of 512 .. 2048 floats, produced slice by slice by the stage's
// workgroups and needed whole by every workgroup of the next stage (an all-gather), 66 times per token.  Between launches the
// boundary costs ~4.4 us (profiles/r05_prior_sampling_kernel_trace.txt).  Here G persistent workgroups (one per CU) run T
// stages of
//     write my slice of row[t & 1] with write-through stores (sc0 sc1: the bytes leave the non-coherent L2)
//     s_waitcnt vmcnt(0); one relaxed agent-scope atomic add on a ticket
//     poll the ticket with relaxed agent-scope loads until it reads G (t + 1)          (no fence anywhere)
//     read the WHOLE row back with sc0 sc1 loads (L2 / L1 bypassed) and fold it into a checksum the next slice depends on
// and the host divides the launch time by T.  Variants: `fence` = the textbook form (plain stores, __threadfence() before the
// ticket and after the poll, plain loads), `ticket only` = the barrier without any payload.  Every run checks the checksum
// chain against the host's own (a stale row shows as a mismatch).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stage_barrier_probe.hip -o tools/probes/stage_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float4 *p, float4 v) {
  const f32x4 r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ f32x4 load_wt(const float4 *p) {     // (the caller waits before it touches the result)
  f32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
  return r;
}

// mode 0: write-through stores / loads, no fence; 1: plain stores / loads + __threadfence(); 2: ticket only (no payload)
template <int MODE>
__global__ __launch_bounds__(256) void stage_chain(float *rows, unsigned *ticket, float *out, int row_floats, int T) {
  const int G = gridDim.x, g = blockIdx.x, tid = threadIdx.x;
  const int slice = row_floats / G;                 // floats of the row this workgroup produces (>= 4)
  const int nq = row_floats / 4;
  __shared__ float red[256];
  float carry = 1.0f;                               // what the previous row folded to
  for (int t = 0; t < T; ++t) {
    float *row = rows + (size_t)(t & 1) * row_floats;
    if (MODE != 2) {
      // my slice: values depend on the previous row's fold (a stale read breaks the chain)
      for (int i = tid * 4; i < slice; i += 256 * 4) {
        const int e = g * slice + i;
        const float4 v = make_float4(carry + 1e-3f * (e % 97), carry - 1e-3f * ((e + 1) % 89), 0.5f * carry + 1e-3f * ((e + 2) % 83),
                                     0.25f * carry - 1e-3f * ((e + 3) % 79));
        if (MODE == 0) store_wt(reinterpret_cast<float4 *>(row + e), v);
        else *reinterpret_cast<float4 *>(row + e) = v;
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (MODE == 1) __threadfence();
      __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)G * (unsigned)(t + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) break;            // (bounded: a lost wake-up must not hang the box)
      }
      if (MODE == 1) __threadfence();
    }
    __syncthreads();
    if (MODE != 2) {
      float s = 0.f;
      f32x4 v[8];                                   // up to 32 KB rows: all of a thread's loads in flight, one wait
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int qd = min(tid + 256 * k, nq - 1);  // (clamped: every load is issued, the sum skips the duplicates)
        if (MODE == 0) v[k] = load_wt(reinterpret_cast<const float4 *>(row) + qd);
        else { const float4 t4 = reinterpret_cast<const float4 *>(row)[qd]; v[k] = f32x4{t4.x, t4.y, t4.z, t4.w}; }
      }
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; ++k) if (tid + 256 * k < nq) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      red[tid] = s;
      __syncthreads();
      for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
      }
      carry = red[0] / (float)row_floats;           // every workgroup folds the same row: the same carry everywhere
      __syncthreads();
    }
  }
  if (tid == 0) out[g] = carry;
}

static float host_chain(int row_floats, int T) {
  float carry = 1.0f;
  std::vector<float> row(row_floats);
  for (int t = 0; t < T; ++t) {
    for (int e = 0; e < row_floats; e += 4) {
      row[e] = carry + 1e-3f * (e % 97); row[e + 1] = carry - 1e-3f * ((e + 1) % 89);
      row[e + 2] = 0.5f * carry + 1e-3f * ((e + 2) % 83); row[e + 3] = 0.25f * carry - 1e-3f * ((e + 3) % 79);
    }
    double s = 0;
    for (float v : row) s += v;
    carry = (float)(s / row_floats);
  }
  return carry;
}

int main() {
  float *rows, *out;
  unsigned *ticket;
  hipMalloc(&rows, 2 * 8192 * sizeof(float));
  hipMalloc(&out, 1024 * sizeof(float));
  hipMalloc(&ticket, 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int T = 200;
  printf("stage boundary inside one launch, us per stage (T = %d stages, one 256-thread workgroup per CU):\n", T);
  printf("%-6s %-10s %-28s %-28s %-14s\n", "G", "row", "write-through, no fence", "plain + __threadfence", "ticket only");
  for (int G : {16, 64, 128, 256}) {
    for (int row_floats : {1024, 2048, 8192}) {       // 4 KB (d_model 512 x 2 ... ), 8 KB (feed-forward 2048), 32 KB
      if (row_floats / G < 4) continue;
      char cell[3][64];
      for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f, got = 0.f;
        bool ok = true;
        for (int rep = 0; rep < 5; ++rep) {
          hipMemset(ticket, 0, 256);
          hipMemset(rows, 0, 2 * 8192 * sizeof(float));
          hipDeviceSynchronize();
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(stage_chain<0>, dim3(G), dim3(256), 0, 0, rows, ticket, out, row_floats, T);
          else if (mode == 1) hipLaunchKernelGGL(stage_chain<1>, dim3(G), dim3(256), 0, 0, rows, ticket, out, row_floats, T);
          else hipLaunchKernelGGL(stage_chain<2>, dim3(G), dim3(256), 0, 0, rows, ticket, out, row_floats, T);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms = 0;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
          std::vector<float> h(G);
          hipMemcpy(h.data(), out, G * sizeof(float), hipMemcpyDeviceToHost);
          if (mode != 2) {
            const float want = host_chain(row_floats, T);
            for (float v : h) if (!(fabsf(v - want) <= 1e-3f * fabsf(want) + 1e-4f)) ok = false;
            got = h[0];
          }
        }
        (void)got;
        snprintf(cell[mode], sizeof cell[mode], "%6.2f%s", best * 1e3f / T, ok ? "" : "  STALE / WRONG");
      }
      printf("%-6d %-10d %-28s %-28s %-14s\n", G, row_floats * 4, cell[0], cell[1], cell[2]);
    }
  }
  return 0;
}
