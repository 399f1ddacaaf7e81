// debug: dump the LDS-resident HCA tables
#include "../vgaudio_amd/csrc/hca_device.hpp"
#include <cstdio>
using namespace vga::hca;
__global__ void k(LdsTables* out){ __shared__ LdsTables T; load_tables(T, threadIdx.x, 256); __syncthreads();
  unsigned char* s=(unsigned char*)&T; unsigned char* d=(unsigned char*)out; for(int i=threadIdx.x;i<(int)sizeof(LdsTables);i+=256) d[i]=s[i]; }
int main(){ LdsTables* d; hipMalloc(&d,sizeof(LdsTables)); k<<<1,256>>>(d); LdsTables h; hipMemcpy(&h,d,sizeof h,hipMemcpyDeviceToHost);
  printf("sizeof %zu\n", sizeof(LdsTables));
  printf("sin[0]=%.17g sin[1]=%.17g sin[127]=%.17g sin[190]=%.17g cos[127]=%.17g\n",h.sin_t[0],h.sin_t[1],h.sin_t[127],h.sin_t[190],h.cos_t[127]);
  printf("window[0]=%.17g [127]=%.17g\n",h.window[0],h.window[127]);
  printf("deq[0]=%.17g deq[63]=%.17g qs[63]=%.17g inv[1]=%.17g step[1]=%.17g dz[8]=%.17g dz[15]=%.17g\n",h.dequant_scale[0],h.dequant_scale[63],h.quant_scale[63],h.inv_step[1],h.step[1],h.dead_zone[8],h.dead_zone[15]);
  printf("shuffle %d %d %d %d\n",h.shuffle[0],h.shuffle[1],h.shuffle[2],h.shuffle[127]);
  printf("enc_bits[1]: "); for(int i=0;i<16;i++) printf("%d ",h.enc_bits[1][i]); printf("\nenc_value[3]: "); for(int i=0;i<16;i++) printf("%d ",h.enc_value[3][i]);
  printf("\ndec_bits[2]: "); for(int i=0;i<16;i++) printf("%d ",h.dec_bits[2][i]); printf("\ndec_value[2]: "); for(int i=0;i<16;i++) printf("%d ",h.dec_value[2][i]);
  printf("\nmax_bits: "); for(int i=0;i<16;i++) printf("%d ",h.max_bits[i]); printf("\nres_curve: "); for(int i=0;i<64;i++) printf("%d ",h.res_curve[i]); printf("\n"); return 0; }
