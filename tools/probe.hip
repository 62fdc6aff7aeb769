// GPU probe (tools/, not product): device limits, HBM copy ceilings, MALL chunk experiment, strided-tile
// access bandwidth.  Informs tile widths / chunking in the planner.  Prints JSON lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template<typename V> __global__ void k_inplace(V* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { V v = p[i]; v.x = v.x * 1.0001f; p[i] = v; }
}
template<typename V> __global__ void k_copy(const V* __restrict__ a, V* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { b[i] = a[i]; }
}
// tile access: each WG handles a tile of ROWS rows x SEG bytes; rows separated by stride bytes; in place.
// lanes: seg/8 lanes across the segment (float2), remaining lanes across rows.
__global__ void k_tile(float2* p, int rows, int segElems, size_t strideElems, size_t tilesPerRowBlock, size_t rowBlockElems) {
    size_t tile = blockIdx.x; size_t rb = tile / tilesPerRowBlock, tc = tile % tilesPerRowBlock;
    float2* base = p + rb * rowBlockElems + tc * segElems;
    int lanesPerRow = segElems; int rowsPerIter = blockDim.x / lanesPerRow;
    int c = threadIdx.x % lanesPerRow, r0 = threadIdx.x / lanesPerRow;
    float2 v[16];
    for (int r = r0; r < rows; r += rowsPerIter * 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) { int rr = r + j * rowsPerIter; if (rr < rows) v[j] = base[(size_t)rr * strideElems + c]; }
#pragma unroll
        for (int j = 0; j < 16; j++) { int rr = r + j * rowsPerIter; if (rr < rows) { v[j].x *= 1.0001f; base[(size_t)rr * strideElems + c] = v[j]; } }
    }
}
static float timeit(hipStream_t s, int iters, const std::function<void()>& f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); hipStreamSynchronize(s);
    hipEventRecord(a, s); for (int i = 0; i < iters; i++) f(); hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / iters;
}
#include <functional>
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int n; hipGetDeviceCount(&n);
    int maxDyn = 0; hipDeviceGetAttribute(&maxDyn, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"arch\":\"%s\",\"ndev\":%d,\"CUs\":%d,\"sharedPerBlock\":%zu,\"attrMaxShared\":%d,\"sharedPerMP\":%zu,\"l2\":%d,\"maxThreads\":%d,\"grid\":[%d,%d,%d],\"regsPerBlock\":%d,\"warp\":%d,\"clockMHz\":%d,\"memGB\":%.1f}\n",
        p.name, p.gcnArchName, n, p.multiProcessorCount, p.sharedMemPerBlock, maxDyn, p.maxSharedMemoryPerMultiProcessor, p.l2CacheSize, p.maxThreadsPerBlock,
        p.maxGridSize[0], p.maxGridSize[1], p.maxGridSize[2], p.regsPerBlock, p.warpSize, p.clockRate / 1000, p.totalGlobalMem / 1e9);
    hipStream_t s; hipStreamCreate(&s);
    size_t bytes = 1ull << 30; void *A, *T; CK(hipMalloc(&A, bytes)); CK(hipMalloc(&T, bytes));
    hipMemset(A, 0, bytes); hipMemset(T, 0, bytes);
    { std::vector<float> h(1 << 20); for (auto& x : h) x = rand() / (float)RAND_MAX; for (size_t o = 0; o < bytes; o += 4 << 20) hipMemcpy((char*)A + o, h.data(), 4 << 20, hipMemcpyHostToDevice); }
    for (int grid : {2048, 8192, 65536}) {
        float ms = timeit(s, 10, [&] { hipLaunchKernelGGL(k_inplace<float4>, dim3(grid), dim3(256), 0, s, (float4*)A, bytes / 16); });
        printf("{\"probe\":\"inplace_f4\",\"grid\":%d,\"ms\":%.4f,\"alg_GBps\":%.1f}\n", grid, ms, 2.0 * bytes / ms / 1e6);
        ms = timeit(s, 10, [&] { hipLaunchKernelGGL(k_inplace<float2>, dim3(grid), dim3(256), 0, s, (float2*)A, bytes / 8); });
        printf("{\"probe\":\"inplace_f2\",\"grid\":%d,\"ms\":%.4f,\"alg_GBps\":%.1f}\n", grid, ms, 2.0 * bytes / ms / 1e6);
        ms = timeit(s, 10, [&] { hipLaunchKernelGGL(k_copy<float4>, dim3(grid), dim3(256), 0, s, (const float4*)A, (float4*)T, bytes / 16); });
        printf("{\"probe\":\"copy_f4\",\"grid\":%d,\"ms\":%.4f,\"alg_GBps\":%.1f}\n", grid, ms, 2.0 * bytes / ms / 1e6);
    }
    // MALL chunk experiment: A[chunk] -> T[0:chunk] -> A[chunk]; algorithmic traffic counted as 2*bytes (one read + one write of A)
    for (size_t ch : {4ull << 20, 16ull << 20, 32ull << 20, 64ull << 20, 128ull << 20, 256ull << 20, 1024ull << 20}) {
        float ms = timeit(s, 5, [&] {
            for (size_t o = 0; o < bytes; o += ch) {
                hipLaunchKernelGGL(k_copy<float4>, dim3(4096), dim3(256), 0, s, (const float4*)((char*)A + o), (float4*)T, ch / 16);
                hipLaunchKernelGGL(k_copy<float4>, dim3(4096), dim3(256), 0, s, (const float4*)T, (float4*)((char*)A + o), ch / 16);
            }
        });
        printf("{\"probe\":\"mall_pingpong\",\"chunkMiB\":%zu,\"ms\":%.4f,\"alg_GBps\":%.1f,\"actual_GBps\":%.1f}\n", ch >> 20, ms, 2.0 * bytes / ms / 1e6, 4.0 * bytes / ms / 1e6);
    }
    // strided tile experiment: emulate four-step column pass on a batch of [rows x cols] matrices (c32)
    for (int logN : {16, 20, 22}) {
      for (int logRows : {7, 8, 9}) {
        for (int seg : {8, 16, 32, 64}) {
            int rows = 1 << logRows; size_t cols = (1ull << logN) / rows; if ((size_t)seg > cols) continue;
            size_t N = 1ull << logN; size_t nmat = (bytes / 8) / N;
            size_t tilesPerMat = cols / seg; size_t ntiles = nmat * tilesPerMat;
            int threads = 256;
            float ms = timeit(s, 5, [&] { hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(threads), 0, s, (float2*)A, rows, seg, cols, tilesPerMat, N); });
            printf("{\"probe\":\"tile_inplace\",\"log2N\":%d,\"rows\":%d,\"segBytes\":%d,\"ms\":%.4f,\"alg_GBps\":%.1f}\n", logN, rows, seg * 8, ms, 2.0 * bytes / ms / 1e6);
        }
      }
    }
    return 0;
}
