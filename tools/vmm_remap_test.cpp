// vmm_remap_test.cpp -- after hipMemUnmap + hipMemMap of ANOTHER handle at the same virtual address, do kernels reach the
// new physical memory?  (Every composition that tools/vmm_interleave_probe.cpp and tools/vmm_piece_probe.cpp mapped into
// one re-used address range timed the same, and fe_split_alloc's probe -- candidates mapped one after the other at one
// nursery slot -- saw one class across the whole memory: a stale translation would explain both.)
// Handles A and B; A mapped at V and filled with 1.0, unmapped; B mapped at V (or, control, at a fresh address) and filled
// with 2.0; then both are read back through addresses that were NEVER used before (bump allocation inside one big
// reservation -- hipMemAddressReserve hands a freed range out again).  Expected: A = 1.0, B = 2.0.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/vmm_remap_test.cpp -o build/vmm_remap_test
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
static const size_t MIB = 1ull << 20, GIB = 1ull << 30;
__global__ void fill(double* p, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static hipMemAllocationProp prop;
static hipMemAccessDesc acc, none;
static char* arena;
static size_t bump = 0;
static char* fresh(size_t size) { char* p = arena + bump; bump += (size + 2 * MIB - 1) / (2 * MIB) * (2 * MIB); return p; }
static void map_at(char* va, hipMemGenericAllocationHandle_t h, size_t size) {
    CK(hipMemMap(va, size, 0, h, 0));
    CK(hipMemSetAccess(va, size, &acc, 1));
}
static void read2(hipMemGenericAllocationHandle_t h, size_t size, double* first, double* last) {   // through a never-used address
    char* v = fresh(size);
    map_at(v, h, size);
    CK(hipMemcpy(first, v, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(last, v + size - 8, 8, hipMemcpyDeviceToHost));
    CK(hipMemUnmap(v, size));
}
int main() {
    CK(hipSetDevice(0));
    prop = hipMemAllocationProp{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    acc = hipMemAccessDesc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    none = acc;
    none.flags = hipMemAccessFlagsProtNone;
    CK(hipMemAddressReserve((void**)&arena, 512 * GIB, 2 * MIB, nullptr, 0));
    const char* modes[] = {"B at a FRESH address (control)", "unmap, map B at the same address", "unmap, hipDeviceSynchronize, map B at the same address",
                           "set access NONE, unmap, map B at the same address", "unmap, 200 ms sleep + sync, map B at the same address"};
    for (size_t size : {2 * MIB, 128 * MIB, 1024 * MIB}) {
        for (int mode = 0; mode < 5; ++mode) {
            hipMemGenericAllocationHandle_t A, B;
            CK(hipMemCreate(&A, size, &prop, 0));
            CK(hipMemCreate(&B, size, &prop, 0));
            const size_t n = size / 8;
            {   // known contents through never-used addresses
                char* t = fresh(size);
                map_at(t, A, size);
                fill<<<1024, 256>>>((double*)t, n, -1.0);
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(t, size));
                t = fresh(size);
                map_at(t, B, size);
                fill<<<1024, 256>>>((double*)t, n, -2.0);
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(t, size));
            }
            char* V = fresh(size);
            map_at(V, A, size);
            fill<<<1024, 256>>>((double*)V, n, 1.0);
            CK(hipDeviceSynchronize());
            if (mode == 3) {
                hipError_t e = hipMemSetAccess(V, size, &none, 1);
                if (e != hipSuccess) { printf("(hipMemSetAccess NONE: %s) ", hipGetErrorString(e)); (void)hipGetLastError(); }
            }
            CK(hipMemUnmap(V, size));
            if (mode == 2) CK(hipDeviceSynchronize());
            if (mode == 4) { hipDeviceSynchronize(); struct timespec ts = {0, 200000000}; nanosleep(&ts, nullptr); hipDeviceSynchronize(); }
            char* VB = mode == 0 ? fresh(size) : V;
            map_at(VB, B, size);
            fill<<<1024, 256>>>((double*)VB, n, 2.0);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(VB, size));
            double a0, a1, b0, b1;
            read2(A, size, &a0, &a1);
            read2(B, size, &b0, &b1);
            printf("size %5zu MiB, %-58s A = %4.1f / %4.1f (want 1.0)   B = %4.1f / %4.1f (want 2.0)   %s\n", size / MIB, modes[mode], a0, a1, b0, b1,
                   (a0 == 1.0 && a1 == 1.0 && b0 == 2.0 && b1 == 2.0) ? "OK" : "WRONG PHYSICAL MEMORY");
            CK(hipMemRelease(A));
            CK(hipMemRelease(B));
        }
    }
    return 0;
}
