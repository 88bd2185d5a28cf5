#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
int main(int argc,char**argv){
  size_t n=(size_t)6<<30; int huge=argc>1&&atoi(argv[1]);
  double t0=now(); char*p=malloc(n);
  if(huge) { size_t a=((size_t)p+((size_t)2<<20)-1)&~(((size_t)2<<20)-1); int r=madvise((void*)a, n-(a-(size_t)p)-((size_t)2<<20), MADV_HUGEPAGE); if(r) perror("madvise"); }
  double t1=now(); for(size_t i=0;i<n;i+=4096)p[i]=1; double t2=now(); memset(p,2,n); double t3=now(); free(p); double t4=now();
  printf("huge=%d touch %.3f s, memset %.3f s, free %.3f s\n",huge,t2-t1,t3-t2,t4-t3); return 0; }
