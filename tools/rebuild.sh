#!/bin/bash
# Rebuild the product library (hipcc, gfx950) and the CPU emulation of the same kernel sources (g++, tests only).
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$root/self-supervised-mvs_amd/csrc" 2>&1 | grep -E "error|Error" || true
make -C "$root/tests/cpu_emul" 2>&1 | grep -E " error|Error" || true
ls -la "$root/self-supervised-mvs_amd/libmvs_hip.so" "$root/tests/cpu_emul/libmvs_emul.so"
