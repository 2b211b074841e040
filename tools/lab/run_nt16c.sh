#!/bin/bash
cd "$(dirname "$0")/../.."
for b in noload_nolds c_nosplit c_half c_nosplit_half c_nostore c_all; do
  timeout -k 5 60 tools/bin/nt16_$b 160984 138 138 138 $b || exit 1
  timeout -k 5 60 tools/bin/nt16_$b 321968 138 138 138 $b || exit 1
done
