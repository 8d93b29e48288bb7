// Which XCD does workgroup i of a 1-D grid run on?  Prints, per (blockIdx.x & 7), the histogram of HW_REG_XCC_ID.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/xcd_probe.hip -o tools/probe/xcd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out, int spin) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xf;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);      // keep blocks resident for a while
}
int main() {
    for (int nblk : {512, 4096, 4104, 20000, 512, 4096}) {
        unsigned* d; hipMalloc(&d, nblk * 4);
        static int round = 0;
        if (++round > 4) {      // a grid that is NOT a multiple of 8 first: where does the next grid's workgroup 0 go?
            unsigned* e; hipMalloc(&e, 64 * 4);
            hipLaunchKernelGGL(probe, dim3(13), dim3(256), 65536, 0, e, 1);
        }
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 65536, 0, d, 200);
        unsigned* h = (unsigned*)malloc(nblk * 4);
        hipMemcpy(h, d, nblk * 4, hipMemcpyDeviceToHost);
        int hist[8][16] = {};
        for (int i = 0; i < nblk; ++i) hist[i & 7][h[i]]++;
        int per[16] = {};
        for (int i = 0; i < nblk; ++i) per[h[i]]++;
        printf("grid %d: workgroup 0 on xcc%u; per XCD:", nblk, h[0]);
        for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
        printf("\n");
        for (int k = 0; k < 8; ++k) { printf("  bid&7=%d:", k); for (int x = 0; x < 16; ++x) if (hist[k][x]) printf(" xcc%d x%d", x, hist[k][x]); printf("\n"); }
        hipFree(d); free(h);
    }
    return 0;
}
