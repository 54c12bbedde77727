"""Option sets that pin est-fact under NON-default options (the reference's 24 options, src/options.ggo:48-370,
checked and copied by src/configuration.c:45-176): shared by the CPU test (host logic over the oracle) and the
`-m gpu` test (the product binary).  Every option is off its default in at least one set; see the table in
tests/golden/README.md.

A set = (id, command line, config.ini text or None, name of the ini file, expected effective options).  The
expected options are what the test hands to the compiled reference (oracle/_ref/est-fact-core reads
`ref-options.ini`, oracle/ref_core_driver.c) -- written from THIS table, not from the product's config-dump.ini,
so the precedence command line > configuration file > default (src/configuration.c:257-277) is checked too."""

DEFAULTS = {
    "min-factor-length": "15", "min-intron-length": "40", "max-intron-length": "0", "min-string-depth-rate": "0.2",
    "max-prefix-discarded-rate": "0.60", "max-suffix-discarded-rate": "0.60", "max-prefix-discarded": "50",
    "max-suffix-discarded": "50", "min-distance-of-splice-sites": "50", "max-no-of-factorizations": "0",
    "max-difference-of-coverage": "0.05", "max-difference-of-no-of-exons": "5", "max-difference-of-gap-length": "20",
    "complexity-threshold": "20.0", "retain-externals": "true", "max-pairings-in-CMEG": "80",
    "max-shortest-pairing-frequence": "0.4", "suff-pref-length-intron": "70", "suff-pref-length-est": "30",
    "suff-pref-length-genomic": "30", "max-single-factorization-time": "900",
}
FLAGS = ("no-transitive-reduction", "no-short-edge-compaction")

# (id, argv, ini text, ini file name, effective non-default options)
SETS = [
    ("factor18-cli", ["--min-factor-length=18"], None, None, {"min-factor-length": "18"}),
    ("factor12-short-opts", ["-l", "12", "-d", "0.35", "-B", "25"], None, None,
     {"min-factor-length": "12", "min-string-depth-rate": "0.35", "min-intron-length": "25"}),
    ("no-externals", ["--retain-externals=false"], None, None, {"retain-externals": "false"}),
    ("no-externals-short", ["-E", "false", "-D", "10"], None, None,
     {"retain-externals": "false", "min-distance-of-splice-sites": "10"}),
    ("windows", ["--suff-pref-length-intron=40", "--suff-pref-length-est=20", "--suff-pref-length-genomic=45"], None, None,
     {"suff-pref-length-intron": "40", "suff-pref-length-est": "20", "suff-pref-length-genomic": "45"}),
    ("windows-wide", ["--suff-pref-length-intron=90", "--suff-pref-length-est=45", "--suff-pref-length-genomic=25"], None, None,
     {"suff-pref-length-intron": "90", "suff-pref-length-est": "45", "suff-pref-length-genomic": "25"}),
    ("cmeg", ["--max-pairings-in-CMEG=6", "--max-shortest-pairing-frequence=0.1"], None, None,
     {"max-pairings-in-CMEG": "6", "max-shortest-pairing-frequence": "0.1"}),
    ("no-reduction", ["--no-transitive-reduction"], None, None, {"no-transitive-reduction": None}),
    ("no-compaction", ["--no-short-edge-compaction", "--max-intron-length=3000"], None, None,
     {"no-short-edge-compaction": None, "max-intron-length": "3000"}),
    ("neither", ["--no-transitive-reduction", "--no-short-edge-compaction"], None, None,
     {"no-transitive-reduction": None, "no-short-edge-compaction": None}),
    ("discarded", ["-p", "0.2", "-s", "0.3", "-P", "10", "-S", "5"], None, None,
     {"max-prefix-discarded-rate": "0.2", "max-suffix-discarded-rate": "0.3", "max-prefix-discarded": "10", "max-suffix-discarded": "5"}),
    ("filters", ["--max-no-of-factorizations=1", "--max-difference-of-coverage=0.5", "--max-difference-of-no-of-exons=0",
                 "--max-difference-of-gap-length=2"], None, None,
     {"max-no-of-factorizations": "1", "max-difference-of-coverage": "0.5", "max-difference-of-no-of-exons": "0",
      "max-difference-of-gap-length": "2"}),
    ("filters-off", ["--max-difference-of-gap-length=-1", "--max-difference-of-no-of-exons=-1", "--complexity-threshold=4.5"], None, None,
     {"max-difference-of-gap-length": "-1", "max-difference-of-no-of-exons": "-1", "complexity-threshold": "4.5"}),
    ("time", ["--max-single-factorization-time=1200", "--min-intron-length=0"], None, None,
     {"max-single-factorization-time": "1200", "min-intron-length": "0"}),
    # config.ini in the current directory (the default --config-file), gengetopt's file format
    ("ini", [], "min-factor-length = 17\nretain-externals = false\n# a comment\nno-transitive-reduction\nsuff-pref-length-est=\"25\"\n",
     "config.ini", {"min-factor-length": "17", "retain-externals": "false", "no-transitive-reduction": None, "suff-pref-length-est": "25"}),
    # the command line wins over the file; what the file alone says stays
    ("cli-over-ini", ["--min-factor-length=20", "-E", "true"],
     "min-factor-length = 17\nretain-externals = false\nmin-intron-length = 60\n", "config.ini",
     {"min-factor-length": "20", "min-intron-length": "60"}),
    # a configuration file under another name
    ("config-file", ["--config-file=other.ini", "-D", "5"], "max-pairings-in-CMEG = 10\nmin-distance-of-splice-sites = 30\n", "other.ini",
     {"max-pairings-in-CMEG": "10", "min-distance-of-splice-sites": "5"}),
]


def ref_options_text(effective):
    """ref-options.ini for oracle/_ref/est-fact-core: the effective configuration in the reference's own
    config-dump.ini format (name="value", a flag as its bare name)."""
    lines = []
    for k, v in DEFAULTS.items():
        lines.append('%s="%s"' % (k, effective.get(k, v)))
    for f in FLAGS:
        if f in effective:
            lines.append(f)
    return "\n".join(lines) + "\n"


def covered_options():
    got = set()
    for _id, _argv, _ini, ini_name, eff in SETS:
        got.update(eff)
        if ini_name and ini_name != "config.ini":
            got.add("config-file")
    return got
