import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_f2" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
print("k_f2 launches", len(d), "durations us:", [round(x,1) for x in d])
