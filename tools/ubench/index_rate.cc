#include "../../jn_cuclark_amd/host/input.hpp"
#include "../../jn_cuclark_amd/host/reads.hpp"
#include <chrono>
#include <cstdio>
#include <omp.h>
using namespace std::chrono;
int main(int argc,char**argv){ int nt=atoi(argv[2]); omp_set_num_threads(nt);
 std::string err; host::InputImage img; img.load(argv[1],err);
 const uint8_t*t=img.data(); size_t nb=img.size();
 for(int rep=0;rep<3;rep++){
  auto t0=steady_clock::now();
  // touch pass: sum bytes in parallel
  unsigned long long sum=0;
  #pragma omp parallel for reduction(+:sum) schedule(static)
  for(size_t i=0;i<nb;i+=4096) sum+=t[i];
  auto t1=steady_clock::now();
  std::vector<host::ReadIndex> part(nt);
  #pragma omp parallel for schedule(static,1)
  for(int p=0;p<nt;p++){ size_t a=host::record_start_at_or_after(t,nb,nb/nt*p,true), b=p+1==nt?nb:host::record_start_at_or_after(t,nb,nb/nt*(p+1),true); std::string e; host::index_reads(t+a,b-a,part[p],e);} 
  auto t2=steady_clock::now();
  printf("touch %.1f ms, thread-index %.1f ms (sum %llu)\n",duration<double,std::milli>(t1-t0).count(),duration<double,std::milli>(t2-t1).count(),sum);
 }
}
