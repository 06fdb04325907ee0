#!/bin/bash
# builds a variant of the library for A/B runs on one box: ab_build.sh <tag> [extra compiler flags]; -> instantvnr_amd/ab/libvnr_amd_<tag>.so
R=/root/repo
tag=$1; shift
mkdir -p $R/instantvnr_amd/ab
make -C $R/instantvnr_amd/csrc -j8 -s BUILD=build_$tag OUT=../ab/libvnr_amd_$tag.so EXTRA="$*" 2>&1 | grep -v "^$" | grep -v warning | head -20
ls -la $R/instantvnr_amd/ab/libvnr_amd_$tag.so
# what it was built from: a stale variant library once cost two GPU calls
echo "[ab_build] $tag: HEAD $(git -C $R rev-parse --short HEAD)$(git -C $R diff --quiet -- instantvnr_amd/csrc || echo '+dirty'), csrc $(cat $R/instantvnr_amd/csrc/*.hip $R/instantvnr_amd/csrc/*.h $R/instantvnr_amd/csrc/*.cpp | md5sum | cut -c1-12), flags '$*'" | tee $R/instantvnr_amd/ab/libvnr_amd_$tag.txt
