#!/bin/bash
# tools/build_variants.sh name1:"cnn flags"[:"geometry flags"] ...: builds tools/_variants/libmsi_<name>.so from the current sources (the cnn*.hip units -- and geometry.hip when a third
# field is given -- recompiled per variant with the flags, in parallel; the other objects of the installed build are reused).  Variants travel to the GPU box with the snapshot.
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/_variants /tmp/vbuild
python -m matryodshka_amd.build > /dev/null || exit 1
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imatryodshka_amd/csrc -Wno-unused-function"
BASE_CNN=$(python -c "from matryodshka_amd import build; print(' '.join(dict(build.SOURCES)['cnn.hip']))")
CNN_UNITS=$(python -c "from matryodshka_amd import build; print(' '.join(u[:-4] for u in build.CNN_UNITS))")
BASE_GEO=$(python -c "from matryodshka_amd import build; print(' '.join(dict(build.SOURCES)['geometry.hip']))")
pids=()
for spec in "$@"; do
  IFS=':' read -r name flags gflags <<< "$spec"
  ( geo=matryodshka_amd/csrc/_obj/geometry.o
    if [ -n "$gflags" ]; then
      [ "$gflags" == "-" ] && gflags=""
      /opt/rocm/bin/hipcc $COMMON $BASE_GEO $gflags -c matryodshka_amd/csrc/geometry.hip -o /tmp/vbuild/geo_$name.o 2>/tmp/vbuild/$name.glog || { echo "FAILED geometry $name"; exit 1; }
      geo=/tmp/vbuild/geo_$name.o
    fi
    objs=""; ok=1
    for u in $CNN_UNITS; do   # (the convolution path's translation units: matryodshka_amd/build.py CNN_UNITS)
      /opt/rocm/bin/hipcc $COMMON $BASE_CNN $flags -c matryodshka_amd/csrc/$u.hip -o /tmp/vbuild/${u}_$name.o 2>>/tmp/vbuild/$name.log || ok=0
      objs="$objs /tmp/vbuild/${u}_$name.o"
    done
    [ $ok == 1 ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_variants/libmsi_$name.so matryodshka_amd/csrc/_obj/common.o matryodshka_amd/csrc/_obj/probe.o $geo $objs &&
    echo "built $name (cnn: $BASE_CNN $flags; geometry: $BASE_GEO $gflags)" || { echo "FAILED $name"; tail -5 /tmp/vbuild/$name.log; } ) &
  pids+=($!)
done
wait "${pids[@]}"
