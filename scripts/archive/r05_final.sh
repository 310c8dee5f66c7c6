#!/bin/bash
cd "$(dirname "$0")/.."
bash scripts/r05_fullsuite.sh
bash scripts/r05_evidence.sh
