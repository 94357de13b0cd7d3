#!/bin/bash
# an alternative libmdx.so with extra -D flags for mdx_rdf.hip (other objects from the last
# regular build): bash scratch/build_alt.sh out.so -DMDX_TIMING_SWITCHES ...
cd "$(dirname "$0")/.."
out=${1:?output}; shift
mkdir -p "$(dirname "$out")"
B=lammps_analysis_amd/build
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wall -Wno-unused-function -I include -I lammps_analysis_amd/csrc "$@" \
  -c lammps_analysis_amd/csrc/mdx_rdf.hip -o /tmp/mdx_rdf_alt.o || exit 1
objs=$(ls $B/*.o | grep -v mdx_rdf)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" /tmp/mdx_rdf_alt.o $objs -ldl
