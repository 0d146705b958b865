#!/bin/bash
# Round 3, second validation pass at HEAD: profiles (stats + PMC) of the headline step, full `-m gpu` suite, smoke, bench lines
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
tools/r03_prof.sh pmc > gpurun_out/r03_prof2.log 2>&1; tail -22 gpurun_out/r03_prof2.log
tools/r03_final.sh
