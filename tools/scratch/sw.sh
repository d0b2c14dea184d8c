for v in rs256 rs512; do echo "== $v"; CSRK_LIBRARY=$PWD/csr_amd/libcsrk_$v.so tools/kstats_rowops.sh ks_$v 2>&1 | grep -v "^E2026\|^W2026" | grep "headline\|c1_kernel\|c3_kernel"; done
