#!/bin/bash
# tools/sweep_cmd.sh VAR "v1 v2 ..." <command...> -- runs <command> once per value of the environment variable VAR (experiment helper)
var=$1; vals=$2; shift 2
for v in $vals; do echo "$var=$v: $(env "$var=$v" timeout 300 "$@" 2>/dev/null | cut -c1-220)"; done
