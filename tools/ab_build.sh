#!/bin/bash
# builds a variant of the library for A/B runs on one box: ab_build.sh <tag> [extra compiler flags]; -> instantvnr_amd/ab/libvnr_amd_<tag>.so
R=/root/repo
tag=$1; shift
mkdir -p $R/instantvnr_amd/ab
make -C $R/instantvnr_amd/csrc -j8 -s BUILD=build_$tag OUT=../ab/libvnr_amd_$tag.so EXTRA="$*" 2>&1 | grep -v "^$" | grep -v warning | head -20
ls -la $R/instantvnr_amd/ab/libvnr_amd_$tag.so
