cd "$GRAFT_REPO_ROOT"
for k in 0 1 2 3 4 5 6 7; do echo "== phase $k"; THESIA_AMD_LIB=scripts/variants/libthesia_amd_subw_prof$k.so python3 scripts/subwave_prof.py 2>&1 | tail -1; done
