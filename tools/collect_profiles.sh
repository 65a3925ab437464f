#!/bin/bash
# Copies the summaries of tools/prof_round.sh and tools/prof_soup.sh (gpurun_out/prof_r03, gpurun_out/prof_soup_r03)
# into profiles/ under the round's names.  Usage: tools/collect_profiles.sh     (repository root, after the GPU call)
set -e
A=gpurun_out/prof_r03; B=gpurun_out/prof_soup_r03
cp $A/kernel_stats.txt profiles/r03_kernel_stats.txt
cp $A/pmc_summary.json profiles/r03_pmc_summary.json
cp $A/pmc_traffic.json profiles/pmc_traffic.json
cp $A/bench_line.json profiles/r03_bench_line.json
cp $B/kernel_stats.txt profiles/r03_soup_kernel_stats.txt
cp $B/pmc_summary.json profiles/r03_soup_pmc_summary.json
cp $B/pmc_traffic_soup.json profiles/pmc_traffic_soup.json
cp $B/bench_line.json profiles/r03_soup_bench_line_profiled.json
