// tests/emul/icc_harness.cpp — TEST-ONLY: the product's ICC command-language decoder (jxl_coder_amd/csrc/host_icc.inc, static functions)
// compiled into a stand-alone AddressSanitizer binary.  stdin: the "encoded ICC" byte string (what the entropy stage hands to
// icc_unpredict); exit code 0 = decoded, 1 = rejected cleanly; an out-of-bounds access aborts with ASan's report (exit != 0, 1).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../jxl_coder_amd/csrc/host_bits.h"
#include "../../jxl_coder_amd/csrc/host_icc.inc"
int main() {
  std::vector<uint8_t> enc; uint8_t buf[4096]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, stdin)) > 0) enc.insert(enc.end(), buf, buf + n);
  std::vector<uint8_t> out;
  const int rc = icc_unpredict(enc, &out);
  printf("%d %zu\n", rc, out.size());
  (void)read_icc_stream;
  return rc ? 1 : 0;
}
