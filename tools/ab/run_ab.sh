#!/bin/bash
# A/B of two builds of the library on one box: tools/ab/libnr_engine_old.so (a build of an earlier source state) against the in-tree one.
# usage: tools/ab/run_ab.sh <command ...>   -- runs the command with the new build, then with the old one swapped in
set -e
echo "=== NEW build"; "$@"
cp news_recommendation_amd/libnr_engine.so /tmp/libnr_engine_new.so
cp tools/ab/libnr_engine_old.so news_recommendation_amd/libnr_engine.so
echo "=== OLD build"; "$@" || true
cp /tmp/libnr_engine_new.so news_recommendation_amd/libnr_engine.so
echo "=== NEW build again"; "$@"
