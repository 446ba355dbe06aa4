#!/bin/bash
# GPU box: is the slow first pass of the CLI over a freshly written clip a property of the file (tmpfs pages read for the first time) or of
# the CLI?  A fresh 4.8-GB file per line, read twice: dd (one thread), then tools/microbench/read_first_pass with 1 / 4 / 8 threads and
# pread, memcpy out of a mapping, memcpy after MADV_POPULATE_READ.
cd "$(dirname "$0")/.."
fresh() { python3 -c "
import os
blob = os.urandom(3110406)
with open('/dev/shm/tm_first_pass.bin', 'wb') as f:
    for i in range(1536): f.write(blob)
"; }
fresh; for i in 1 2; do dd if=/dev/shm/tm_first_pass.bin of=/dev/null bs=8M 2>&1 | tail -1; done
for mode in pread mmap populate; do for nt in 1 4 8; do fresh; tools/microbench/read_first_pass /dev/shm/tm_first_pass.bin $nt $mode 2; done; done
rm -f /dev/shm/tm_first_pass.bin
