set pagination off
set confirm off
set breakpoint pending on
set height 0
break cdkf_custom_kernel
run
delete 1
break *(&cdkf_custom_kernel + 0xe20)
commands
silent
printf "SEEDS p=%d  (p==2)v15=%#x (p==0)v17=%#x (p==1)v19=%#x  v211=%#x s[2:3]=%#x:%#x\n", $v28[0], $v15[0], $v17[0], $v19[0], $v211[0], $s3, $s2
continue
end
continue
quit
