"""profiles/traffic_latest.json from the raw TCC counter means of tools/pmc_traffic.sh (gpurun_out/traffic_<tag>_raw.json).
    python tools/traffic_latest.py <round> <raw json of configs[2]> <raw json of configs[1]> [<raw json of the configs[3] slab>]
Units (MI355X_MICROARCH.md, checked in the same run on kernels of known traffic and the same access width):
FETCH_SIZE KiB x 2 on gfx950 (sort_count_kernel reads exactly 4 B per particle), WRITE_SIZE KiB (load_maxwellian_kernel
writes exactly 32 B per particle)."""
import json
import sys

DECKS = {2: ("256x256x256 periodic two-stream, 2 species x 64 ppc, dt=0.95 Courant, sort_interval=10", 256 ** 3 * 64),
         1: ("128x128x128 periodic two-stream, 2 species x 32 ppc, dt=0.95 Courant, sort_interval=10", 128 ** 3 * 32)}


def main():
    rnd = int(sys.argv[1])
    out = {}
    for cfg, path in ((2, sys.argv[2]), (1, sys.argv[3])):
        raw = json.load(open(path))
        name, np_ = DECKS[cfg]
        push = [k for k in raw if k.startswith("advance_p_kernel")]
        hist = [k for k in push if k.endswith(", true, false>")]   # the launch before a sort also counts the sort's histogram
        sort = [k for k in push if k.endswith(", false, true>")]   # the launch that sorts as it pushes
        push = [k for k in push if k not in hist and k not in sort]
        assert len(push) == 1, push
        count = [k for k in raw if k.startswith("wg_count_kernel") or k.startswith("sort_count_kernel")][0]   # either reads exactly 4 B per particle
        f = raw[push[0]]["FETCH_SIZE"]["mean"] * 1024 * 2
        w = raw[push[0]]["WRITE_SIZE"]["mean"] * 1024
        chk_r = raw[count]["FETCH_SIZE"]["mean"] * 1024 * 2 / (4.0 * np_)
        chk_w = raw["load_maxwellian_kernel"]["WRITE_SIZE"]["mean"] * 1024 / (32.0 * np_)
        out[name] = {"kernel": push[0], "round": rnd, "fetch_bytes_per_launch": int(f), "write_bytes_per_launch": int(w),
                     "hbm_bytes_per_launch": int(f + w),
                     "how": "tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, --kernel-trace only); "
                            "FETCH_SIZE KiB x2 (gfx950 counts 64 B of each 128 B request; checked on the sort's count kernel = 4 B/particle: "
                            "%.4f of expected), WRITE_SIZE KiB x1 (checked on load_maxwellian_kernel = 32 B/particle: %.4f of expected); "
                            "raw: profiles/r%02d_traffic_config%d_raw.json" % (chk_r, chk_w, rnd, cfg)}
        if hist:
            out[name]["histogram_launch"] = {"kernel": hist[0], "fetch_bytes_per_launch": int(raw[hist[0]]["FETCH_SIZE"]["mean"] * 2048),
                                             "write_bytes_per_launch": int(raw[hist[0]]["WRITE_SIZE"]["mean"] * 1024)}
        if sort:
            out[name]["sorting_launch"] = {"kernel": sort[0], "fetch_bytes_per_launch": int(raw[sort[0]]["FETCH_SIZE"]["mean"] * 2048),
                                           "write_bytes_per_launch": int(raw[sort[0]]["WRITE_SIZE"]["mean"] * 1024)}
        print(name, "fetch %.2f GB write %.2f GB" % (f / 1e9, w / 1e9), "checks", round(chk_r, 4), round(chk_w, 4))
    if len(sys.argv) > 4:
        # the configs[3] slab: launches of three instances (charge-0 copies; charged species sorted by cell / by tile only) -- the
        # bench line's roofline averages over all of them, so does this entry; the instances are listed apart
        raw = json.load(open(sys.argv[4]))
        name = ("32x256x128 one x-slab of configs[3] (trecon-part at 256x256x128 over 8 GPUs): periodic x,y / conducting reflecting z, "
                "pair plasma vth=0.6c + its 2 tracer copies, 4 species x 64 ppc, dt=0.95 Courant, sort_interval=-20")
        push = {k: v for k, v in raw.items() if k.startswith("advance_p_kernel")}
        n = sum(v["FETCH_SIZE"]["launches"] for v in push.values())
        f = sum(v["FETCH_SIZE"]["mean"] * 2048 * v["FETCH_SIZE"]["launches"] for v in push.values()) / n
        w = sum(v["WRITE_SIZE"]["mean"] * 1024 * v["WRITE_SIZE"]["launches"] for v in push.values()) / sum(v["WRITE_SIZE"]["launches"] for v in push.values())
        out[name] = {"kernel": "advance_p_kernel (all instances of the deck, launch-weighted)", "round": rnd, "fetch_bytes_per_launch": int(f),
                     "write_bytes_per_launch": int(w), "hbm_bytes_per_launch": int(f + w),
                     "instances": {k: {"launches": v["FETCH_SIZE"]["launches"], "fetch_bytes_per_launch": int(v["FETCH_SIZE"]["mean"] * 2048),
                                       "write_bytes_per_launch": int(v["WRITE_SIZE"]["mean"] * 1024)} for k, v in push.items()},
                     "how": "as above; raw: profiles/r%02d_traffic_config3_slab_raw.json" % rnd}
        print(name[:40], "fetch %.2f GB write %.2f GB" % (f / 1e9, w / 1e9))
    json.dump(out, open("profiles/traffic_latest.json", "w"), indent=1)


main()
