// tests/asan/signal_asan_main.cpp -- TEST INFRASTRUCTURE ONLY: the product's `signal` step (pansvr_amd/csrc/signal_step.h +
// bam_reader.h, host code) built with AddressSanitizer + UBSan, so malformed BAM records can be thrown at it on the CPU.
#include "../../pansvr_amd/csrc/signal_step.h"
int main(int argc, char **argv) { return psvr::signal_main(argc, argv); }
