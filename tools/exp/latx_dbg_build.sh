#!/bin/bash
# Development: the LATX_DBG build of the library for tools/exp/latx_steps.py -- k_millerlatx.hip with its time stamps compiled in,
# linked with the shipped objects of every other unit (run `make` in bgls_amd/csrc first).  Output: tools/exp/libbgls_hip_latxdbg.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/hip_dbg
(cd bgls_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLATX_DBG -I../../include -c k_millerlatx.hip -o ../../build/hip_dbg/k_millerlatx.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libbgls_hip_latxdbg.so $(ls build/hip/*.o | grep -v k_millerlatx.o) build/hip_dbg/k_millerlatx.o
ls -la tools/exp/libbgls_hip_latxdbg.so
