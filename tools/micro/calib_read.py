import pandas as pd, glob, sys
for d in sys.argv[1:]:
    f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0]
    df=pd.read_csv(f)
    df["k"]=df.Kernel_Name.str.split("(").str[0]
    print(df.groupby(["k","Counter_Name"]).Counter_Value.mean().to_string())
