#!/bin/bash
# builds the TEST-ONLY host twin of the solver core (see hostsim.cpp header)
set -e
cd "$(dirname "$0")"
if [ ! -f libhostsim.so ] || [ hostsim.cpp -nt libhostsim.so ] || [ ../../myriad_amd/csrc/hs_solver.h -nt libhostsim.so ] || [ ../../myriad_amd/csrc/systems_gen.h -nt libhostsim.so ] || [ ../../myriad_amd/csrc/rollout.h -nt libhostsim.so ] || [ ../../myriad_amd/csrc/os_solver.h -nt libhostsim.so ]; then
  g++ -O2 -std=c++17 -fopenmp -fPIC -shared hostsim.cpp -o libhostsim.so
fi
