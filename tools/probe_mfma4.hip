// Probe: lane mapping + issue rate of v_mfma_f32_4x4x4_16b_bf16 on gfx950, and ds_read_b64_tr_b16 mapping.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstring>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); u += 0x7FFF + ((u>>16)&1); return (uint16_t)(u>>16);} 

__global__ void k_map(f32x4* o, const s16x4* a, const s16x4* b){
  f32x4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  o[threadIdx.x]=c;
}
constexpr int ITERS=2048;
__global__ void k_rate4(f32x4* o, const s16x4* a, const s16x4* b){
  s16x4 av=a[threadIdx.x&63], bv=b[threadIdx.x&63];
  f32x4 c[8];
  for(int i=0;i<8;++i) c[i]=f32x4{0,0,0,0};
  for(int it=0;it<ITERS;++it){
#pragma unroll
    for(int i=0;i<8;++i) c[i]=__builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av,bv,c[i],0,0,0);
  }
  f32x4 s={0,0,0,0}; for(int i=0;i<8;++i) s+=c[i];
  o[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_rate4_dep(f32x4* o, const s16x4* a, const s16x4* b){   // one dependent chain
  s16x4 av=a[threadIdx.x&63], bv=b[threadIdx.x&63];
  f32x4 c={0,0,0,0};
  for(int it=0;it<ITERS;++it){
#pragma unroll
    for(int i=0;i<8;++i) c=__builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av,bv,c,0,0,0);
  }
  o[blockIdx.x*blockDim.x+threadIdx.x]=c;
}
__global__ void k_rate32(f32x4* o, const s16x8* a, const s16x8* b){
  s16x8 av=a[threadIdx.x&63], bv=b[threadIdx.x&63];
  f32x16 c[4];
  for(int i=0;i<4;++i) for(int j=0;j<16;++j) c[i][j]=0;
  for(int it=0;it<ITERS;++it){
#pragma unroll
    for(int i=0;i<4;++i) c[i]=__builtin_amdgcn_mfma_f32_32x32x16_bf16(av,bv,c[i],0,0,0);
  }
  f32x4 s={0,0,0,0}; for(int i=0;i<4;++i) {s[0]+=c[i][0]; s[1]+=c[i][5]; s[2]+=c[i][10]; s[3]+=c[i][15];}
  o[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_rate16(f32x4* o, const s16x8* a, const s16x8* b){
  s16x8 av=a[threadIdx.x&63], bv=b[threadIdx.x&63];
  f32x4 c[8];
  for(int i=0;i<8;++i) c[i]=f32x4{0,0,0,0};
  for(int it=0;it<ITERS;++it){
#pragma unroll
    for(int i=0;i<8;++i) c[i]=__builtin_amdgcn_mfma_f32_16x16x32_bf16(av,bv,c[i],0,0,0);
  }
  f32x4 s={0,0,0,0}; for(int i=0;i<8;++i) s+=c[i];
  o[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
// tr read probe: LDS holds M[row][col] = row*256+col (16-bit), 64 rows x 64 cols (128 B rows).
__global__ void k_tr(short* o){
  __shared__ __attribute__((aligned(16))) short lds[64*64];
  for(int i=threadIdx.x;i<64*64;i+=64) lds[i]=(short)(((i/64)<<8)|(i%64));
  __syncthreads();
  int lane=threadIdx.x; int g=lane>>4, l16=lane&15; int q=l16>>2, p=l16&3;
  // block for group g: rows 4g..4g+3, cols 0..15 : lane 4q+p supplies address of row q, cols 4p..4p+3
  int addr=((4*g+q)*64 + 4*p)*2;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + addr));
  for(int j=0;j<4;++j) o[lane*4+j]=v[j];
}

template<typename K, typename A> int rate(const char* name, K kern, A* a, A* b, f32x4* o, double macs_per_instr, int per_iter){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int w: {1,2}){
    int grid=256*w;
    hipLaunchKernelGGL(kern,dim3(grid),dim3(256),0,0,o,a,b); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for(int r=0;r<10;++r) hipLaunchKernelGGL(kern,dim3(grid),dim3(256),0,0,o,a,b);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=10;
    double instr_per_simd=(double)w*ITERS*per_iter;
    double cyc=ms*1e-3*2.4e9/instr_per_simd;
    double tmacs=(double)grid*4*ITERS*per_iter*macs_per_instr/ms*1e-9;
    printf("%-22s waves/SIMD=%d %8.3f ms  ~%.2f cyc/instr/SIMD@2.4GHz  %.1f TMAC/s chip\n",name,w,ms,cyc,tmacs);
  }
  return 0;
}

int main(){
  // ---- mapping probe ----
  std::vector<uint16_t> ha(64*4), hb(64*4); std::vector<float> fa(64*4), fb(64*4);
  for(int l=0;l<64;++l) for(int k=0;k<4;++k){ fa[l*4+k]=(float)((l*7+k*3)%13-6); fb[l*4+k]=(float)((l*5+k*11)%17-8); ha[l*4+k]=f2bf(fa[l*4+k]); hb[l*4+k]=f2bf(fb[l*4+k]); }
  s16x4 *da,*db; f32x4* dout; CK(hipMalloc(&da,64*8)); CK(hipMalloc(&db,64*8)); CK(hipMalloc(&dout,256*2*256*16));
  CK(hipMemcpy(da,ha.data(),64*8,hipMemcpyHostToDevice)); CK(hipMemcpy(db,hb.data(),64*8,hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_map,dim3(1),dim3(64),0,0,dout,da,db); CK(hipDeviceSynchronize());
  std::vector<float> ho(64*4); CK(hipMemcpy(ho.data(),dout,64*16,hipMemcpyDeviceToHost));
  // hypothesis: block=lane/4; A row i = lane%4, B col j = lane%4; D[i=reg][j=lane%4]
  int bad=0;
  for(int l=0;l<64;++l) for(int r=0;r<4;++r){ int blk=l/4, j=l%4; float ref=0; for(int k=0;k<4;++k) ref+=fa[(blk*4+r)*4+k]*fb[(blk*4+j)*4+k]; if(std::fabs(ref-ho[l*4+r])>1e-3) {++bad; if(bad<5) printf("mismatch lane %d reg %d got %f ref %f\n",l,r,ho[l*4+r],ref);} }
  printf("4x4x4_16b mapping hypothesis (blk=lane/4, A row=lane%%4, B col=lane%%4, D[reg][lane%%4]): %s (%d bad)\n", bad?"FAIL":"OK", bad);
  // ---- tr read probe ----
  short* dtr; CK(hipMalloc(&dtr,64*4*2)); hipLaunchKernelGGL(k_tr,dim3(1),dim3(64),0,0,dtr); CK(hipDeviceSynchronize());
  std::vector<short> htr(256); CK(hipMemcpy(htr.data(),dtr,512,hipMemcpyDeviceToHost));
  int badt=0; for(int l=0;l<64;++l) for(int j=0;j<4;++j){ int g=l>>4,i=l&15; int exp=((4*g+j)<<8)|i; if(htr[l*4+j]!=exp){++badt; if(badt<6) printf("tr lane %d elem %d got row %d col %d\n",l,j,(htr[l*4+j]>>8)&255,htr[l*4+j]&255);} }
  printf("ds_read_b64_tr_b16 hypothesis (lane i of group gets column i, element q = row q): %s\n", badt?"FAIL":"OK");
  // ---- rates ----
  s16x8 *da8,*db8; CK(hipMalloc(&da8,64*16)); CK(hipMalloc(&db8,64*16)); CK(hipMemset(da8,0x3c,64*16)); CK(hipMemset(db8,0x3c,64*16));
  rate("mfma_4x4x4_16b_bf16", k_rate4, da, db, dout, 16*64.0, 8);
  rate("mfma_4x4x4 dep-chain", k_rate4_dep, da, db, dout, 16*64.0, 8);
  rate("mfma_32x32x16_bf16", k_rate32, da8, db8, dout, 32*32*16.0, 4);
  rate("mfma_16x16x32_bf16", k_rate16, da8, db8, dout, 16*16*32.0, 8);
  return 0;
}
