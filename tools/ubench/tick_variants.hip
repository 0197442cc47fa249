// Scratch microbenchmark (not product): streaming variants of the tick-mode saw kernel
// and a plain copy ceiling with the same 8 B read + 4 B write per element.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define uint4 u32x4
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template<int UNROLL, bool NT, int BS>
__global__ __launch_bounds__(BS) void tick_k(const uint4* __restrict__ inc, const uint4* __restrict__ si, uint4* __restrict__ so, int32_t* bus, uint32_t ngroups)
{
    int32_t acc = 0;
    const uint32_t stride = gridDim.x * BS;
    uint32_t g = blockIdx.x * BS + threadIdx.x;
    for (; g + (UNROLL-1)*stride < ngroups; g += UNROLL*stride) {
        uint4 a[UNROLL], b[UNROLL];
#pragma unroll
        for (int u=0;u<UNROLL;u++) {
            if (NT) { a[u] = __builtin_nontemporal_load(&inc[g+u*stride]); b[u] = __builtin_nontemporal_load(&si[g+u*stride]); }
            else    { a[u] = inc[g+u*stride]; b[u] = si[g+u*stride]; }
        }
#pragma unroll
        for (int u=0;u<UNROLL;u++) {
            uint4 o; o.x=b[u].x+a[u].x; o.y=b[u].y+a[u].y; o.z=b[u].z+a[u].z; o.w=b[u].w+a[u].w;
            acc += (a[u].x? (int32_t)b[u].x>>4:0) + (a[u].y? (int32_t)b[u].y>>4:0) + (a[u].z? (int32_t)b[u].z>>4:0) + (a[u].w? (int32_t)b[u].w>>4:0);
            if (NT) __builtin_nontemporal_store(o, &so[g+u*stride]); else so[g+u*stride]=o;
        }
    }
    for (; g < ngroups; g += stride) {
        uint4 a=inc[g], b=si[g]; uint4 o; o.x=b.x+a.x; o.y=b.y+a.y; o.z=b.z+a.z; o.w=b.w+a.w;
        acc += (a.x? (int32_t)b.x>>4:0) + (a.y? (int32_t)b.y>>4:0) + (a.z? (int32_t)b.z>>4:0) + (a.w? (int32_t)b.w>>4:0);
        so[g]=o;
    }
    // wave reduce + block reduce
    for (int o=32;o>0;o>>=1) acc += __shfl_xor(acc,o);
    __shared__ int32_t w[16];
    if ((threadIdx.x&63)==0) w[threadIdx.x>>6]=acc;
    __syncthreads();
    if (threadIdx.x==0) { int32_t t=0; for (int i=0;i<BS/64;i++) t+=w[i]; atomicAdd(bus, t); }
}

__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, uint32_t ngroups)
{
    const uint32_t stride = gridDim.x*256u;
    for (uint32_t g = blockIdx.x*256u+threadIdx.x; g<ngroups; g+=stride) { uint4 x=a[g], y=b[g]; uint4 r; r.x=x.x^y.x; r.y=x.y^y.y; r.z=x.z^y.z; r.w=x.w^y.w; o[g]=r; }
}

template<typename F> float timeit(F f, int reps){ hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); for(int i=0;i<3;i++) f(); hipDeviceSynchronize(); hipEventRecord(e0); for(int i=0;i<reps;i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); return ms/reps; }

int main(int argc,char**argv){
    const uint32_t n = argc>1? (uint32_t)atol(argv[1]) : (1u<<26);
    const uint32_t ng = n/4;
    uint32_t *inc,*s0,*s1; int32_t* bus;
    CK(hipMalloc(&inc,(size_t)n*4)); CK(hipMalloc(&s0,(size_t)n*4)); CK(hipMalloc(&s1,(size_t)n*4)); CK(hipMalloc(&bus,256));
    std::vector<uint32_t> h(n); for(uint32_t i=0;i<n;i++) h[i]=i*2654435761u|1;
    CK(hipMemcpy(inc,h.data(),(size_t)n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(s0,h.data(),(size_t)n*4,hipMemcpyHostToDevice));
    const double bytes = 12.0*n;
    int grids[] = {128,192,256,320,384,448,512,640,768,1024};
    for (int gx : grids) {
        float a = timeit([&]{ hipLaunchKernelGGL((tick_k<1,true,256>),dim3(gx),dim3(256),0,0,(const uint4*)inc,(const uint4*)s0,(uint4*)s1,bus,ng); },20);
        float b = timeit([&]{ hipLaunchKernelGGL((tick_k<1,true,512>),dim3(gx),dim3(512),0,0,(const uint4*)inc,(const uint4*)s0,(uint4*)s1,bus,ng); },20);
        float c = timeit([&]{ hipLaunchKernelGGL((tick_k<1,true,1024>),dim3(gx),dim3(1024),0,0,(const uint4*)inc,(const uint4*)s0,(uint4*)s1,bus,ng); },20);
        float d = timeit([&]{ hipLaunchKernelGGL((tick_k<2,true,1024>),dim3(gx),dim3(1024),0,0,(const uint4*)inc,(const uint4*)s0,(uint4*)s1,bus,ng); },20);
        float e = timeit([&]{ hipLaunchKernelGGL((tick_k<1,false,1024>),dim3(gx),dim3(1024),0,0,(const uint4*)inc,(const uint4*)s0,(uint4*)s1,bus,ng); },20);
        printf("grid %6d: nt bs256 %.1f bs512 %.1f bs1024 %.1f bs1024u2 %.1f | plain bs1024 %.1f GB/s\n", gx, bytes/a/1e6, bytes/b/1e6, bytes/c/1e6, bytes/d/1e6, bytes/e/1e6);
    }
    return 0;
}
