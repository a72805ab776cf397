// which device sqrt forms are correctly rounded on gfx950? (exactness matters for parity)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* x, float* o, int n){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return; float v=x[i];
  o[i*4+0]=sqrtf(v); o[i*4+1]=__fsqrt_rn(v); o[i*4+2]=__builtin_sqrtf(v); o[i*4+3]=(float)sqrt((double)v); }
int main(){ const int n=1<<20; std::vector<float> x(n), o(4*n); unsigned s=1; for(int i=0;i<n;++i){ s=s*1664525u+1013904223u; x[i]=(float)(s>>8)*(1.0f/16777216.0f)*((i&1)?1000.f:1.f)+1e-6f; }
  float *dx,*dout; hipMalloc(&dx,n*4); hipMalloc(&dout,n*16); hipMemcpy(dx,x.data(),n*4,hipMemcpyHostToDevice);
  k<<<n/256,256>>>(dx,dout,n); hipMemcpy(o.data(),dout,n*16,hipMemcpyDeviceToHost);
  int bad[4]={0,0,0,0}; for(int i=0;i<n;++i){ float ref=(float)std::sqrt((double)x[i]); for(int j=0;j<4;++j) if(o[i*4+j]!=ref) bad[j]++; }
  printf("mismatches of %d: sqrtf %d, __fsqrt_rn %d, __builtin_sqrtf %d, (float)sqrt(double) %d\n", n,bad[0],bad[1],bad[2],bad[3]); return 0; }
