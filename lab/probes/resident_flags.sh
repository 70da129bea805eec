for f in 0 128 256 384 512 1024 1536; do echo "== v4_flags=$f"; timeout 200 python lab/probes/resident_probe.py 4 5000 v4_flags=$f 2>&1 | grep "resident        \|ALL\|MISM"; done
