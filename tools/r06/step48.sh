#!/bin/bash
# Round 6, step 48: which launches of a layer have an idle gap in front of them (batches 2 and 8)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 600 bash tools/r06/decode_prof.sh gap2 2 > /dev/null 2>&1; tail -14 $O/decode_prof_gap2.txt | cut -c1-150
timeout 600 bash tools/r06/decode_prof.sh gap8 8 > /dev/null 2>&1; tail -14 $O/decode_prof_gap8.txt | cut -c1-150
