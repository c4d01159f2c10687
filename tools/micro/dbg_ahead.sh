g++ -std=c++17 -O2 -Iinclude examples/StepLoop_amd.C -Lroot-simple-mcmc_amd/lib -lsmcmc_amd -Wl,-rpath,$PWD/root-simple-mcmc_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/sl.exe
/tmp/sl.exe 5 3 700 1 0 /tmp/t0.csv
/tmp/sl.exe 5 3 700 1 1 /tmp/t1.csv
python3 - <<'PY'
a=open('/tmp/t0.csv').read().splitlines(); b=open('/tmp/t1.csv').read().splitlines()
print(len(a),len(b))
h=a[0].split(',')
for i,(x,y) in enumerate(zip(a,b)):
    if x!=y:
        xs=x.split(','); ys=y.split(',')
        print("first differing entry", i, [(h[k],xs[k],ys[k]) for k in range(len(xs)) if xs[k]!=ys[k]][:8])
        break
PY
