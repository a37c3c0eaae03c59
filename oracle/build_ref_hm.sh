#!/bin/bash
# oracle/_ref: the reference's OWN encoders, compiled from the sources where they lie under
# /root/reference (every needed .cpp directly with g++; the reference's makefiles are not used).
# TEST INFRASTRUCTURE: they are the real callers / consumers on both sides of the path's boundary,
# used by the end-to-end tests (tests/test_gpu_e2e.py) and scripts/; nothing here is shipped.
#
#   oracle/_ref/hm_ai/TAppEncoderUnchanged   HM-16.5_Test_AI as it is: its hook runs
#                                            `python video_to_cu_depth.py <yuv> <w> <h> <qp>` (TAppEncCfg.cpp:2317-2321)
#   oracle/_ref/hm_ai/TAppEncoderInProcess   same sources with tools/hm_inprocess_patch.py applied to the temporary copy:
#                                            no predictor run up front, no cu_depth.dat -- TEncCu::compressCtu hands each
#                                            picture's own luma to tools/hm_inprocess_hook.c -> libethcnn.so
#   oracle/_ref/hm_ldp/TAppEncoderLDP        HM-16.5_Test_LDP as it is: the encoder side of the LDP file handshake
#                                            (TEncGOP.cpp:1463-1503)
#
# Sources are copied to a temporary directory inside oracle/_ref/ and deleted after the build; only
# the binaries stay (oracle/_ref/ is git-ignored and travels to the GPU box with the snapshot).
set -eu
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(dirname "$HERE")"
REF=/root/reference
[ -d "$REF/HM-16.5_Test_AI/source" ] || { echo "no $REF: keeping prebuilt oracle/_ref (if any)"; exit 0; }
FLAGS="-O2 -w -DMSYS_LINUX -D_LARGEFILE64_SOURCE -D_FILE_OFFSET_BITS=64 -DMSYS_UNIX_LARGEFILE -Isource/Lib -Isource/App/TAppEncoder"
compile_all() {  # cwd holds source/ ; objects -> obj/
    mkdir -p obj
    ls source/Lib/TLibCommon/*.cpp source/Lib/TLibEncoder/*.cpp source/Lib/TLibVideoIO/*.cpp source/Lib/TAppCommon/*.cpp \
       source/Lib/libmd5/*.c source/App/TAppEncoder/*.cpp |
      xargs -P 8 -I{} sh -c 'o=obj/$(echo {} | tr "/" "_").o; case {} in *.c) gcc '"$FLAGS"' -c {} -o $o;; *) g++ '"$FLAGS"' -c {} -o $o;; esac'
}

# ---- HM-16.5_Test_AI: unchanged + in-process hook -------------------------------------------
B="$HERE/_ref/hm_ai"; rm -rf "$B"; mkdir -p "$B"; cd "$B"
cp -r "$REF/HM-16.5_Test_AI/source" source; chmod -R u+w source
compile_all
g++ -o TAppEncoderUnchanged obj/*.o -lpthread -ldl
python3 "$REPO/tools/hm_inprocess_patch.py" source
g++ $FLAGS -c source/App/TAppEncoder/TAppEncCfg.cpp -o obj/source_App_TAppEncoder_TAppEncCfg.cpp.o
g++ $FLAGS -c source/Lib/TLibEncoder/TEncCu.cpp -o obj/source_Lib_TLibEncoder_TEncCu.cpp.o
gcc -std=c99 -O2 -D_POSIX_C_SOURCE=200809L -I"$REPO/include" -c "$REPO/tools/hm_inprocess_hook.c" -o obj/hm_inprocess_hook.o
g++ -o TAppEncoderInProcess obj/*.o -L"$REPO/hevc-complexity-reduction_amd/lib" -lethcnn -lpthread -ldl \
    -Wl,-rpath,'$ORIGIN/../../../hevc-complexity-reduction_amd/lib' -Wl,-rpath,/opt/rocm/lib
rm -rf source obj

# ---- HM-16.5_Test_LDP: unchanged ----------------------------------------------------------------
B="$HERE/_ref/hm_ldp"; rm -rf "$B"; mkdir -p "$B"; cd "$B"
cp -r "$REF/HM-16.5_Test_LDP/source" source; chmod -R u+w source
compile_all
g++ -o TAppEncoderLDP obj/*.o -lpthread -ldl
rm -rf source obj
ls -la "$HERE/_ref/hm_ai" "$HERE/_ref/hm_ldp"
