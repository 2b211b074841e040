#!/bin/bash
# where the time of the two-part NT kernel goes at 1.285 M rows: eight queued 161 k launches, A confined to 16 k rows, full A
cd "$(dirname "$0")/../.."
timeout -k 5 120 tools/bin/nt16_never 160984 138 138 138 never || exit 1
REP=8 timeout -k 5 120 tools/bin/nt16_never 160984 138 138 138 never_x8 || exit 1
timeout -k 5 120 tools/bin/nt16_smalla 1285158 138 138 138 smallA || exit 1
timeout -k 5 120 tools/bin/nt16_never 1285158 138 138 138 never || exit 1
timeout -k 5 120 tools/bin/nt16_always 1285158 138 138 138 always || exit 1
