#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
using clk=std::chrono::steady_clock;
static double now(){return std::chrono::duration<double>(clk::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  size_t GB=2; size_t blk=8<<20; std::vector<char> b(blk,'x');
  const char*dir=argc>1?argv[1]:"/tmp";
  char p1[256],p2[256]; snprintf(p1,256,"%s/wt1.bin",dir); snprintf(p2,256,"%s/wt2.bin",dir);
  auto wr=[&](const char*p){FILE*f=fopen(p,"w"); for(size_t i=0;i<(GB<<30)/blk;i++) fwrite(b.data(),1,blk,f); fclose(f);};
  double t=now(); wr(p1); printf("one file fwrite %.2f GB/s\n", GB/(now()-t));
  t=now(); {std::thread a(wr,p1), c(wr,p2); a.join(); c.join();} printf("two files concurrently %.2f GB/s total\n", 2*GB/(now()-t));
  // parallel pwrite same file
  t=now(); {int fd=open(p1,O_WRONLY|O_CREAT|O_TRUNC,0644); std::vector<std::thread> th; int T=8; size_t per=(GB<<30)/T;
    for(int i=0;i<T;i++) th.emplace_back([&,i]{for(size_t o=0;o<per;o+=blk) pwrite(fd,b.data(),blk,i*per+o);}); for(auto&x:th)x.join(); close(fd);} printf("8 threads pwrite one file %.2f GB/s\n", GB/(now()-t));
  t=now(); {int fd=open(p1,O_RDWR|O_CREAT|O_TRUNC,0644); ftruncate(fd,GB<<30); char*m=(char*)mmap(0,GB<<30,PROT_WRITE,MAP_SHARED,fd,0); std::vector<std::thread> th; int T=8; size_t per=(GB<<30)/T;
    for(int i=0;i<T;i++) th.emplace_back([&,i]{for(size_t o=0;o<per;o+=blk) memcpy(m+i*per+o,b.data(),blk);}); for(auto&x:th)x.join(); munmap(m,GB<<30); close(fd);} printf("8 threads mmap one file %.2f GB/s\n", GB/(now()-t));
  unlink(p1); unlink(p2);
}
