#!/bin/bash
# Builds tools/kbench (conv check / timing through the C ABI, no Python) next to the library it links.
set -e
cd "$(dirname "$0")/.."
python -m passl_amd.csrc.build >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/kbench.cpp -Iinclude -Lpassl_amd/lib -lpassl_hip \
  -Wl,-rpath,'$ORIGIN/../passl_amd/lib' -o tools/kbench
echo tools/kbench
