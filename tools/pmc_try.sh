cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== plain"; python3 $R/tests/gpu_nan_hunt.py 2>&1 | tail -22
echo "== pmc"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcx -o x -- python3 $R/tests/gpu_nan_hunt.py 2>&1 | grep -v "^[EWI]2026" | tail -22
