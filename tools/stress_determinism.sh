#!/bin/bash
# determinism of the matrix-core scan under load over several shapes (tools/determinism_check.py each): key tables, column
# partials and match tables of every round must equal the first round's, word for word
root=$(cd "$(dirname "$0")/.." && pwd)
rc=0
for cfg in "320 70 3 ${R1:-40}" "320 70 24 ${R1:-40}" "1500 200 256 ${R2:-16}" "777 130 64 ${R1:-40}" "2100 33 16 ${R1:-40}" "257 4130 4 ${R1:-40}" "2060 513 6 ${R1:-40}" "64 64 512 ${R1:-40}"; do
  set -- $cfg
  python $root/tools/determinism_check.py --n-orb $1 --n-lbd $2 --pairs $3 --rounds $4 --scan-variant 4 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$cfg', 'key words', d['key_words'], 'diffs: keys', d['key_diffs'], 'partials', d['partial_diffs'], 'tables', d['table_diffs'])
sys.exit(1 if d['key_diffs'] or d['partial_diffs'] or d['table_diffs'] else 0)" || rc=1
done
exit $rc
