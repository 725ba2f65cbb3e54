#!/bin/bash
# Development tool: which GPU box is this? (host, card serial / VBIOS / firmware, partition mode) next to one kernel number
hostname; cat /proc/sys/kernel/random/boot_id 2>/dev/null
rocm-smi --showserial --showvbios --showuniqueid --showcomputepartition --showmemorypartition 2>&1 | grep -E "GPU\[" | head -12
rocm-smi --showfwinfo 2>&1 | grep -E "SMC|MEC |SDMA|PSP|RLC " | head -8
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Waves|LDS|Wavefront|Cacheline|Max Clock" | sed -n 1,40p | sort | uniq -c | sort -rn | head -12
