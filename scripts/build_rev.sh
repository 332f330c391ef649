# Build the library of another git revision beside the product library (for same-box A/B runs through $VQ_AMD_LIB):
#   bash scripts/build_rev.sh <rev> <name>   ->  video-quierer_amd/lib/libvq_amd_<name>.so
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/vqrev.XXXXXX)
git -C "$root" archive "$rev" video-quierer_amd/csrc include scripts/gen_gemm_asm.py scripts/check_asm256.py | tar -x -C "$tmp"
make -C "$tmp/video-quierer_amd/csrc" OUT="$root/video-quierer_amd/lib/libvq_amd_$name.so" OBJDIR="$tmp/obj" > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
rm -rf "$tmp"
ls -la "$root/video-quierer_amd/lib/libvq_amd_$name.so"
