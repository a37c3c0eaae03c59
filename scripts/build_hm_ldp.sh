#!/bin/bash
# Builds, HERE (needs /root/reference), the reference's HM-16.5_Test_LDP encoder UNCHANGED from its
# own sources (every needed .cpp compiled directly with g++; the reference's makefiles are not used)
# into the git-ignored build/hm_ldp/, so that the real encoder side of the LDP file handshake
# (TEncGOP.cpp:1463-1503) can be run against the daemon on the GPU box and against an oracle-backed
# daemon here (scripts/ldp_e2e.py).  Only the binary is kept.
set -eu
REPO="$(cd "$(dirname "$0")/.." && pwd)"
REF=/root/reference/HM-16.5_Test_LDP
B="$REPO/build/hm_ldp"
rm -rf "$B"; mkdir -p "$B/obj"
cp -r "$REF/source" "$B/source"; chmod -R u+w "$B/source"
cd "$B"
SRCS=$(ls source/Lib/TLibCommon/*.cpp source/Lib/TLibEncoder/*.cpp source/Lib/TLibVideoIO/*.cpp source/Lib/TAppCommon/*.cpp source/Lib/libmd5/*.c source/App/TAppEncoder/*.cpp)
FLAGS="-O2 -w -DMSYS_LINUX -D_LARGEFILE64_SOURCE -D_FILE_OFFSET_BITS=64 -DMSYS_UNIX_LARGEFILE -Isource/Lib -Isource/App/TAppEncoder"
echo "$SRCS" | tr ' ' '\n' | xargs -P 8 -I{} sh -c 'o=obj/$(echo {} | tr "/" "_").o; case {} in *.c) gcc '"$FLAGS"' -c {} -o $o;; *) g++ '"$FLAGS"' -c {} -o $o;; esac'
g++ -o TAppEncoderLDP obj/*.o -lpthread -ldl
rm -rf "$B/source" "$B/obj"
ls -la "$B"
