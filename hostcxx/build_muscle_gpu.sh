#!/bin/bash
# Builds hostcxx/_build/muscle_gpu: the reference's own `muscle` with the MI355X posterior stage
# linked in as a drop-in. Needs the reference objects that oracle/build_ref.sh
# compiles from the sources where they lie ($MUSCLE_REF_SRC, default /root/reference/src) — nothing
# of the reference is copied into the repo, and the combined (GPL-3.0) binary is git-ignored; it
# travels to the GPU box as a built artefact.
#
#   reference objects  - consflat.o, alnalnsflat.o, alnmsasflat.o, buildpostflat.o, alignpairflat.o, refineflat.o  (MPCFlat::ConsIter,
#                        MPCFlat::AlignAlns, PProg::AlignMSAsFlat, MPCFlat::BuildPost, AlignPairFlat(_SparsePost) and
#                        MPCFlat::RefineIter — each alone in its translation unit in the reference — are ours)
#                      - calcposteriorflat.o's CalcPosterior symbol, weakened with objcopy so that
#                        hostcxx/mpcflat_gpu.cpp's strong definition wins while CalcPostFlat and the
#                        two vestigial virtuals in the same object stay available; likewise one member each of
#                        super7.o, uclust.o, pprog2.o and mpcflat.o (MPCFlat::CalcPosteriors) - see below
#   + hostcxx/mpcflat_gpu.cpp (g++, against the reference headers) + hostcxx/rand_isolate.cpp
#     (-Wl,--wrap=rand) + -lmpcgpu
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
SRC="${MUSCLE_REF_SRC:-/root/reference/src}"
REFOBJ="$ROOT/oracle/_ref/obj"
OUT="$HERE/_build"
if [ ! -d "$SRC" ] || [ ! -d "$REFOBJ" ]; then
  echo "build_muscle_gpu.sh: reference sources/objects not available (run oracle/build_ref.sh where /root/reference exists) - skipping" >&2
  exit 0
fi
mkdir -p "$OUT"
CXXFLAGS="-std=c++17 -O3 -fopenmp -DNDEBUG -pthread -fPIC -w -I$ROOT/oracle/_ref/inc -I$SRC -I$ROOT/include"
g++ $CXXFLAGS -c "$HERE/mpcflat_gpu.cpp" -o "$OUT/mpcflat_gpu.o"
g++ $CXXFLAGS -c "$HERE/rand_isolate.cpp" -o "$OUT/rand_isolate.o"
objcopy --weaken-symbol=_ZN7MPCFlat13CalcPosteriorEj "$REFOBJ/calcposteriorflat.o" "$OUT/calcposteriorflat_weak.o"
# MPCFlat::BuildPost is ours too (hostcxx/mpcflat_gpu.cpp -> mpcgpu_build_post): buildpostflat.o is not linked at all
rm -f "$OUT/buildpostflat_ref.o"
# Super7::IntraAlignShrubs: ours (parallel over device contexts); the rest of super7.o stays
objcopy --weaken-symbol=_ZN6Super716IntraAlignShrubsEv "$REFOBJ/super7.o" "$OUT/super7_weak.o"
# UClust::Search: ours (all word-count hits aligned in one library call); AlignPairFlat / AlignPairFlat_SparsePost: ours
# (alignpairflat.o is not linked)
objcopy --weaken-symbol=_ZN6UClust6SearchEjRNSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEE "$REFOBJ/uclust.o" "$OUT/uclust_weak.o"
# PProg::Run2: ours (independent joins of the guide tree side by side); AlignAndJoin and the rest of pprog2.o stay
objcopy --weaken-symbol=_ZN5PProg4Run2ERKSt6vectorIjSaIjEES4_ "$REFOBJ/pprog2.o" "$OUT/pprog2_weak.o"
# MPCFlat::CalcPosteriors: ours (a plain loop: the work of all pairs happens inside the first CalcPosterior call); the rest of mpcflat.o stays
objcopy --weaken-symbol=_ZN7MPCFlat14CalcPosteriorsEv "$REFOBJ/mpcflat.o" "$OUT/mpcflat_weak.o"
# MPCFlat::ProgressiveAlign: ours (the joins of a guide-tree level in one library call); the reference's stays in the binary as
# MPCFlat_ProgressiveAlign_ref (our fallback calls it); ProgAln, FreeProgMSAs, FreeSparsePosts of progalnflat.o stay as they are
objcopy --redefine-sym _ZN7MPCFlat16ProgressiveAlignEv=MPCFlat_ProgressiveAlign_ref "$REFOBJ/progalnflat.o" "$OUT/progalnflat_weak.o"
OBJS=$(ls "$REFOBJ"/*.o | grep -v -e '/super7\.o$' -e '/uclust\.o$' -e '/alignpairflat\.o$' -e '/pprog2\.o$' -e '/mpcflat\.o$' -e '/progalnflat\.o$' | grep -v -e '/consflat\.o$' -e '/alnalnsflat\.o$' -e '/alnmsasflat\.o$' -e '/calcposteriorflat\.o$' -e '/buildpostflat\.o$' -e '/refineflat\.o$')
# The product links libmpcgpu.so. tests/test_dropin_emu.py re-runs this script with
# MPCGPU_LIBDIR/MPCGPU_LIBNAME pointing at the SIMT-emulator build of the same library sources
# (tests/emu, test infrastructure) to check the host-side plumbing of this file without a GPU.
LIBDIR="${MPCGPU_LIBDIR:-$ROOT/muscle_amd/csrc}"
LIBNAME="${MPCGPU_LIBNAME:-mpcgpu}"
BIN="${MPCGPU_BIN:-muscle_gpu}"
# --wrap=rand: the reference's rand() (refineflat.cpp:14) gets a private copy of glibc's default
# stream; the HIP runtime in the same process otherwise consumes it (hostcxx/rand_isolate.cpp)
g++ -O3 -fopenmp -pthread -Wl,--wrap=rand $OBJS "$OUT/calcposteriorflat_weak.o" "$OUT/super7_weak.o" "$OUT/uclust_weak.o" "$OUT/pprog2_weak.o" "$OUT/mpcflat_weak.o" "$OUT/progalnflat_weak.o" "$OUT/mpcflat_gpu.o" "$OUT/rand_isolate.o" \
  -L"$LIBDIR" -l"$LIBNAME" -Wl,-rpath,"$LIBDIR" -Wl,-rpath,'$ORIGIN/../../muscle_amd/csrc' -Wl,-rpath,/opt/rocm/lib -o "$OUT/$BIN"
echo "built: $OUT/$BIN"
