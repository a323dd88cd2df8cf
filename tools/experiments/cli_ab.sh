brief() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], {k: d[k] for k in ('files', 'wall_s', 'first_rows_after_s', 'seconds_typing', 'exit_s', 'assemblies_per_s_steady')}, d['phases_s'].get('context_ready'), d.get('rows_written_at', '')[:8])
" "$1"; }
grep -E "thp_|compact_stall|compact_fail" /proc/vmstat | tr '\n' ' '; echo
python tools/cli_probe.py --files 192 --repeats 5 --batch 0 --marks 2>&1 | brief 960
python tools/cli_probe.py --files 192 --repeats 5 --batch 0 --marks 2>&1 | brief 960
python tools/cli_probe.py --files 192 --repeats 96 --batch 0 --marks 2>&1 | brief 18432
python tools/cli_probe.py --files 192 --repeats 96 --batch 0 --marks 2>&1 | brief 18432
KAPTIVE_AMD_READ_AHEAD_GB=0 python tools/cli_probe.py --files 192 --repeats 96 --batch 0 --marks 2>&1 | brief 18432-noahead
grep -E "thp_|compact_stall|compact_fail" /proc/vmstat | tr '\n' ' '; echo
