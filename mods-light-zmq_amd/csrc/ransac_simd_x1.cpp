// 1-lane build of ransac_simd.inc (see the Makefile for the -m flags)
#define VW 1
#define NS simd1
#include "ransac_simd.inc"
