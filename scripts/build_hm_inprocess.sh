#!/bin/bash
# Builds, HERE (needs /root/reference), an HM-16.5_Test_AI encoder whose predictor hook calls
# libethcnn.so IN PROCESS instead of system("python video_to_cu_depth.py ...") -- SURVEY.md 8f row 3.
# The reference sources are copied into the git-ignored build/ directory, the three hook lines of
# TAppEncCfg.cpp:2317-2321 are rewritten there by the python step below, and every needed .cpp is
# compiled directly with g++ (the reference's makefiles are not used).  Nothing of the reference
# enters the repository; the resulting binary travels to the GPU box with the snapshot.
set -eu
REPO="$(cd "$(dirname "$0")/.." && pwd)"
REF=/root/reference/HM-16.5_Test_AI
B="$REPO/build/hm_inprocess"
rm -rf "$B"; mkdir -p "$B/obj"
cp -r "$REF/source" "$B/source"
chmod -R u+w "$B/source"
cp "$B/source/App/TAppEncoder/TAppEncCfg.cpp" "$B/TAppEncCfg_unchanged.cpp"
python3 - "$B/source/App/TAppEncoder/TAppEncCfg.cpp" <<'PY'
import re, sys
p = sys.argv[1]
s = open(p, encoding="latin1").read()
pat = re.compile(r'[ \t]*sprintf\(cmd, "python video_to_cu_depth\.py[^\n]*\n[ \t]*printf\("%s\\n", cmd\);\n[ \t]*assert\(system\(cmd\)==0\);\n')
assert len(pat.findall(s)) == 1, "hook site not found"
s = pat.sub('\tassert(ethcnn_hm_predict(m_pchInputFile, m_iSourceWidth, m_iSourceHeight, m_iQP) == 0);\n', s)
s = s.replace('Void TAppEncCfg::xPrintParameter()', 'extern "C" int ethcnn_hm_predict(const char*, int, int, int);\nVoid TAppEncCfg::xPrintParameter()', 1)
open(p, "w", encoding="latin1").write(s)
print("hook rewritten in", p)
PY
cd "$B"
SRCS=$(ls source/Lib/TLibCommon/*.cpp source/Lib/TLibEncoder/*.cpp source/Lib/TLibVideoIO/*.cpp source/Lib/TAppCommon/*.cpp source/Lib/libmd5/*.c source/App/TAppEncoder/*.cpp)
FLAGS="-O2 -w -DMSYS_LINUX -D_LARGEFILE64_SOURCE -D_FILE_OFFSET_BITS=64 -DMSYS_UNIX_LARGEFILE -Isource/Lib -Isource/App/TAppEncoder"
echo "$SRCS" | tr ' ' '\n' | xargs -P 8 -I{} sh -c 'o=obj/$(echo {} | tr "/" "_").o; case {} in *.c) gcc '"$FLAGS"' -c {} -o $o;; *) g++ '"$FLAGS"' -c {} -o $o;; esac'
gcc -std=c99 -O2 -D_POSIX_C_SOURCE=200809L -I"$REPO/include" -c "$REPO/tools/hm_inprocess_hook.c" -o obj/hm_inprocess_hook.o
g++ -o TAppEncoderInProcess obj/*.o -L"$REPO/hevc-complexity-reduction_amd/lib" -lethcnn -lpthread -ldl \
    -Wl,-rpath,'$ORIGIN/../../hevc-complexity-reduction_amd/lib' -Wl,-rpath,/opt/rocm/lib
# the same objects with the UNCHANGED TAppEncCfg.cpp: the reference's encoder as is (its hook runs
# `python video_to_cu_depth.py ...`), for the full drop-in run on the GPU box
cp TAppEncCfg_unchanged.cpp source/App/TAppEncoder/TAppEncCfg.cpp
g++ $FLAGS -c source/App/TAppEncoder/TAppEncCfg.cpp -o obj/source_App_TAppEncoder_TAppEncCfg.cpp.o
rm -f obj/hm_inprocess_hook.o
g++ -o TAppEncoderUnchanged obj/*.o -lpthread -ldl
rm -rf "$B/source" "$B/obj" "$B/TAppEncCfg_unchanged.cpp"     # keep only the binaries
ls -la "$B"
