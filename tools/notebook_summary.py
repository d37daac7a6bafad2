import sys, json
txt = open(sys.argv[1]).read()
d = json.loads(txt.split("NOTEBOOK_JSON ")[1])
for k, v in d.items():
    print(k, round(v["value"], 1), v["unit"], v["epochs"], v.get("fit_stats"), "ref", round(v["reference"]["value"], 2))
