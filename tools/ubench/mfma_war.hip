// Does a VALU write of an in-flight v_mfma_f32_16x16x32_bf16's SOURCE registers (A, B or C) corrupt the MFMA on gfx950?
// All registers explicit: A = v[40:43], B = v[44:47], acc = v[60:63].  A = B = 1.0 (bf16) so every D element must be 32 per
// MFMA; right after the MFMA (gap = G wait states) the operand under test is overwritten with zeros by four v_mov_b32.
// Prints, per operand and gap, how many of 64 lanes x 4 values x iterations came out wrong.
// build: hipcc -O3 --offload-arch=gfx950 mfma_war.hip -o mfma_war ; run on an MI355X
#include <hip/hip_runtime.h>
#include <stdio.h>

#define SETUP "v_mov_b32 v40, 0x3f803f80\n v_mov_b32 v41, 0x3f803f80\n v_mov_b32 v42, 0x3f803f80\n v_mov_b32 v43, 0x3f803f80\n" \
              "v_mov_b32 v44, 0x3f803f80\n v_mov_b32 v45, 0x3f803f80\n v_mov_b32 v46, 0x3f803f80\n v_mov_b32 v47, 0x3f803f80\n" \
              "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n s_nop 7\n"
#define MFMA "v_mfma_f32_16x16x32_bf16 v[60:63], v[40:43], v[44:47], v[60:63]\n"
#define KILL_A "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
#define KILL_B "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
#define TAIL "s_nop 7\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v60\n v_mov_b32 %1, v61\n v_mov_b32 %2, v62\n v_mov_b32 %3, v63\n"
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v60", "v61", "v62", "v63"

template <int WHICH, int GAP>
__global__ void k(int iters, unsigned *bad) {
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    float d0, d1, d2, d3;
    if (WHICH == 0) {
      if (GAP == 0) asm volatile(SETUP MFMA KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 1) asm volatile(SETUP MFMA "s_nop 0\n" KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 2) asm volatile(SETUP MFMA "s_nop 1\n" KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 4) asm volatile(SETUP MFMA "s_nop 3\n" KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 8) asm volatile(SETUP MFMA "s_nop 7\n" KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
    } else if (WHICH == 1) {
      if (GAP == 0) asm volatile(SETUP MFMA KILL_B TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 1) asm volatile(SETUP MFMA "s_nop 0\n" KILL_B TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 2) asm volatile(SETUP MFMA "s_nop 1\n" KILL_B TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 4) asm volatile(SETUP MFMA "s_nop 3\n" KILL_B TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 8) asm volatile(SETUP MFMA "s_nop 7\n" KILL_B TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
    } else {   // two MFMAs back to back on the same accumulator, then kill A of the SECOND while the first still runs
      if (GAP == 0) asm volatile(SETUP MFMA MFMA KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 1) asm volatile(SETUP MFMA MFMA "s_nop 0\n" KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 2) asm volatile(SETUP MFMA MFMA "s_nop 1\n" KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 4) asm volatile(SETUP MFMA MFMA "s_nop 3\n" KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 8) asm volatile(SETUP MFMA MFMA "s_nop 7\n" KILL_A TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
    }
    if (WHICH == 3) {   // RAW: VALU writes the accumulator (C) then the MFMA reads it, GAP wait states later
#define SET_C "v_mov_b32 v60, 1.0\n v_mov_b32 v61, 1.0\n v_mov_b32 v62, 1.0\n v_mov_b32 v63, 1.0\n"
      if (GAP == 0) asm volatile(SETUP SET_C MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 1) asm volatile(SETUP SET_C "s_nop 0\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 2) asm volatile(SETUP SET_C "s_nop 1\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 4) asm volatile(SETUP SET_C "s_nop 3\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 8) asm volatile(SETUP SET_C "s_nop 7\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
    }
    if (WHICH == 4) {   // RAW: VALU writes A (2.0 in every bf16 slot) then the MFMA reads it
#define SET_A "v_mov_b32 v40, 0x40004000\n v_mov_b32 v41, 0x40004000\n v_mov_b32 v42, 0x40004000\n v_mov_b32 v43, 0x40004000\n"
      if (GAP == 0) asm volatile(SETUP SET_A MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 1) asm volatile(SETUP SET_A "s_nop 0\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 2) asm volatile(SETUP SET_A "s_nop 1\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 4) asm volatile(SETUP SET_A "s_nop 3\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
      if (GAP == 8) asm volatile(SETUP SET_A "s_nop 7\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB);
    }
    if (WHICH == 5) {   // RAW after an MFMA in flight: MFMA, then VALU rewrites the accumulator of a SECOND chain, then MFMA on it
      // (v[64:67] second accumulator): checks VALU write -> MFMA SrcC while the pipe is busy
#define MFMA2 "v_mfma_f32_16x16x32_bf16 v[64:67], v[40:43], v[44:47], v[64:67]\n"
#define SET_C2 "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 1.0\n v_mov_b32 v66, 1.0\n v_mov_b32 v67, 1.0\n"
#define TAIL2 "s_nop 7\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v64\n v_mov_b32 %1, v65\n v_mov_b32 %2, v66\n v_mov_b32 %3, v67\n"
      if (GAP == 0) asm volatile(SETUP MFMA SET_C2 MFMA2 TAIL2 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
      if (GAP == 1) asm volatile(SETUP MFMA SET_C2 "s_nop 0\n" MFMA2 TAIL2 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
      if (GAP == 2) asm volatile(SETUP MFMA SET_C2 "s_nop 1\n" MFMA2 TAIL2 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
      if (GAP == 4) asm volatile(SETUP MFMA SET_C2 "s_nop 3\n" MFMA2 TAIL2 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
      if (GAP == 8) asm volatile(SETUP MFMA SET_C2 "s_nop 7\n" MFMA2 TAIL2 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
    }
    if (WHICH == 6) {   // dependent chain at distance GAP+1: MFMA(acc), GAP other MFMAs on a second accumulator, MFMA(acc) again
      if (GAP == 0) asm volatile(SETUP MFMA MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
      if (GAP == 1) asm volatile(SETUP MFMA MFMA2 MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
      if (GAP == 2) asm volatile(SETUP MFMA MFMA2 MFMA2 MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
      if (GAP == 4) asm volatile(SETUP MFMA "s_nop 1\n" MFMA2 "s_nop 1\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
      if (GAP == 8) asm volatile(SETUP MFMA "v_mov_b32 v64, 0\n s_nop 1\n" MFMA2 "s_nop 1\n" MFMA TAIL : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : : CLOB, "v64", "v65", "v66", "v67");
    }
    const float want = (WHICH == 2 || WHICH == 6) ? 64.0f : (WHICH == 3 || WHICH == 5) ? 33.0f : WHICH == 4 ? 64.0f : 32.0f;
    nbad += (d0 != want) + (d1 != want) + (d2 != want) + (d3 != want);
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int WHICH, int GAP>
void run(unsigned *bad) {
  hipMemset(bad, 0, 4);
  hipLaunchKernelGGL((k<WHICH, GAP>), dim3(512), dim3(512), 0, 0, 200, bad);
  hipDeviceSynchronize();
  unsigned b; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
  const char *nm[] = {"A overwritten", "B overwritten", "A overwritten behind a queued dependent MFMA", "C written by VALU just before",
                      "A written by VALU just before", "C written by VALU just before, pipe busy", "dependent chain (see source for the gap variants)"};
  printf("%-46s gap %d wait states: %u wrong of %u\n", nm[WHICH], GAP, b, 512u * 512u * 200u * 4u);
}

int main() {
  unsigned *bad; hipMalloc(&bad, 4);
  run<0, 0>(bad); run<0, 1>(bad); run<0, 2>(bad); run<0, 4>(bad); run<0, 8>(bad);
  run<1, 0>(bad); run<1, 1>(bad); run<1, 2>(bad); run<1, 4>(bad); run<1, 8>(bad);
  run<2, 0>(bad); run<2, 1>(bad); run<2, 2>(bad); run<2, 4>(bad); run<2, 8>(bad);
  run<3, 0>(bad); run<3, 1>(bad); run<3, 2>(bad); run<3, 4>(bad); run<3, 8>(bad);
  run<4, 0>(bad); run<4, 1>(bad); run<4, 2>(bad); run<4, 4>(bad); run<4, 8>(bad);
  run<5, 0>(bad); run<5, 1>(bad); run<5, 2>(bad); run<5, 4>(bad); run<5, 8>(bad);
  run<6, 0>(bad); run<6, 1>(bad); run<6, 2>(bad); run<6, 4>(bad); run<6, 8>(bad);
  return 0;
}
