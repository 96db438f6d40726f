// 8-lane build of ransac_simd.inc (see the Makefile for the -m flags)
#define VW 8
#define NS simd8
#include "ransac_simd.inc"
