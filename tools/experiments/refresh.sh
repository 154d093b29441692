#!/bin/bash
# Re-base the patches kept here onto the current tree (tests/test_bench_cpu.py: a patch that no longer applies is a claim that no longer holds).
# Each patch is applied with fuzz in a scratch worktree of HEAD and written back as `git diff`; a patch that needs a hand says so and is left alone.
# Commit your work first: the worktree is made from HEAD.
set -e
root=$(git rev-parse --show-toplevel)
for p in "$root"/tools/experiments/*.patch; do
  if git -C "$root" apply --check "$p" 2>/dev/null; then echo "ok       $(basename $p)"; continue; fi
  wt=$(mktemp -d /tmp/qn_patch_wt.XXXXXX); rmdir "$wt"
  git -C "$root" worktree add -q "$wt" HEAD
  if (cd "$wt" && patch -p1 -F3 --no-backup-if-mismatch -s < "$p" && ! find . -name '*.rej' | grep -q .); then
    (cd "$wt" && find . -name '*.orig' -delete && git add -A && git diff --cached) > "$p.new" && mv "$p.new" "$p"
    echo "re-based $(basename $p)"
  else
    echo "NEEDS A HAND: $(basename $p) (rejects in $wt)"; exit 1
  fi
  git -C "$root" worktree remove --force "$wt"
done
