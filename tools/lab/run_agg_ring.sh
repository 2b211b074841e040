#!/bin/bash
cd "$(dirname "$0")/../.."
timeout -k 5 120 tools/bin/agg_ring_lab 160984 138 0 || exit 1
timeout -k 5 120 tools/bin/agg_ring_lab 160984 138 3 || exit 1
