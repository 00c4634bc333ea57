for sw in 2 0; do for ps in 1 0; do
echo "SWEEP=$sw PSUMS=$ps"; GCMI_SWEEP=$sw GCMI_FUSED_PSUMS=$ps GCMI_FUSED_DIAG=1 python bench.py --profile-only --steps 2 --warmup 1 2>&1 | grep fused_bwd | tail -3 | cut -c1-200
GCMI_SWEEP=$sw GCMI_FUSED_PSUMS=$ps python bench.py --profile-only --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
